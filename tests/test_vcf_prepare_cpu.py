"""mfx_vcf_prepare is host work (no index, no device): it runs here.  The prepared run itself is compared with the unprepared one, byte
for byte, on the GPU (tests/test_gpu_variants.py::test_variants_prepared_ahead*)."""
import pytest

from tests import synth


def test_prepare_runs_without_a_device_and_only_once(tmp_path):
    import merfin_amd as m
    k = 21
    names, asm, vcf, read, amers = synth.variant_world(k=k, seed=5)
    vp = str(tmp_path / "in.vcf")
    open(vp, "w").write(vcf)
    for mode in ("filter", "polish"):
        loaded = m.LoadedVcf(vp)
        loaded.prepare(k, mode, names, asm, comb=4)
        with pytest.raises(m.MfxError):
            loaded.prepare(k, mode, names, asm, comb=4)
        loaded.close()
    loaded = m.LoadedVcf(vp)
    with pytest.raises(m.MfxError):
        loaded.prepare(0, "polish", names, asm)                # k out of range
    with pytest.raises(m.MfxError):
        loaded.prepare(65, "polish", names, asm)
    loaded.prepare(31, "polish", names, asm, nosplit=True, debug_path=str(tmp_path / "x"))
    loaded.close()


def test_prepared_run_equals_the_unprepared_one_on_stand_in_values(tmp_path):
    """the host pipeline driven without a device (tools/variants_host_bench.cpp: every path k-mer gets counts from a hash of the text):
    records, log and -debug lines of the prepared run == the unprepared run's, with one cluster per batch and with the default batches"""
    import filecmp
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not shutil.which("g++") or not os.path.exists("/opt/rocm/include"):
        pytest.skip("no g++ / ROCm headers")
    exe = str(tmp_path / "vhb")
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(root, "tools", "variants_host_bench.cpp"), "-I" + os.path.join(root, "merfin_amd", "csrc"),
                           "-I" + os.path.join(root, "include"), "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-L" + os.path.join(root, "merfin_amd"),
                           "-lmerfin_amd", "-Wl,-rpath," + os.path.join(root, "merfin_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    for batch_mb, dev, comb in (("0", "0", "15"), ("64", "1", "15"), ("0", "0", "2")):
        env = dict(os.environ, MFX_VAR_BATCH_MB=batch_mb, TMPDIR=str(tmp_path))
        dbg = [str(tmp_path / "a.dbg"), str(tmp_path / "b.dbg")] if dev == "0" else ["", ""]
        subprocess.check_call([exe, "2e6", "5", dev, dbg[0], comb, "1", "0", ".t_a"], env=env, stdout=subprocess.DEVNULL)
        subprocess.check_call([exe, "2e6", "5", dev, dbg[1], comb, "1", "2", ".t_b"], env=env, stdout=subprocess.DEVNULL)
        try:
            assert filecmp.cmp("/tmp/mfx_vhb.t_a.out.vcf", "/tmp/mfx_vhb.t_b.out.vcf", shallow=False)
            assert filecmp.cmp("/tmp/mfx_vhb.t_a.log", "/tmp/mfx_vhb.t_b.log", shallow=False)
            assert os.path.getsize("/tmp/mfx_vhb.t_a.out.vcf") > 1000
            if dev == "0":
                assert filecmp.cmp(dbg[0], dbg[1], shallow=False)
        finally:
            for t in (".t_a", ".t_b"):
                for ext in (".out.vcf", ".log"):
                    if os.path.exists("/tmp/mfx_vhb" + t + ext):
                        os.remove("/tmp/mfx_vhb" + t + ext)
            if os.path.exists("/tmp/mfx_vhb.vcf"):
                os.remove("/tmp/mfx_vhb.vcf")


def test_shared_traverse_equals_the_host_recursion(tmp_path):
    """mfx_traverse_cluster (csrc/mfx_traverse.h: what the device's traverse kernel runs per cluster) against the host's recursion
    (merfin-variants.C:22-126 restated in mfx_variants.cpp): every eligible cluster of synthetic call sets -- plain, with multi-allelic and
    odd records, damaged -- enumerated both ways (bases, genotype, offset and length rows of every path: MFX_VAR_TRAVERSE_CHECK), and the whole
    run's records and log with the clusters enumerated through the tables == with the recursion"""
    import filecmp
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not shutil.which("g++") or not os.path.exists("/opt/rocm/include"):
        pytest.skip("no g++ / ROCm headers")
    exe = str(tmp_path / "vhb")
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(root, "tools", "variants_host_bench.cpp"), "-I" + os.path.join(root, "merfin_amd", "csrc"),
                           "-I" + os.path.join(root, "include"), "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-L" + os.path.join(root, "merfin_amd"),
                           "-lmerfin_amd", "-Wl,-rpath," + os.path.join(root, "merfin_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    try:
        for mode, comb, varied, damage in (("5", "15", "0", None), ("4", "3", "1", None), ("5", "15", "1", "1"), ("6", "15", "1", None)):
            env = dict(os.environ)
            if damage:
                env["MFX_VHB_DAMAGE"] = damage
            r = subprocess.run([exe, "8e6", mode, "1", "", comb, varied, "0", ".u_c"], env=dict(env, MFX_VAR_TRAVERSE_CHECK="1"), capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-400:]
            checked = [l for l in r.stderr.splitlines() if "traverse check" in l]
            assert checked and int(checked[0].split(":")[1].split()[0]) > 1000, r.stderr[-400:]
            subprocess.check_call([exe, "8e6", mode, "1", "", comb, varied, "0", ".u_a"], env=dict(env, MFX_VAR_DEVICE_TRAVERSE="0"), stdout=subprocess.DEVNULL)
            subprocess.check_call([exe, "8e6", mode, "1", "", comb, varied, "0", ".u_b"], env=dict(env, MFX_VAR_DEVICE_TRAVERSE="1"), stdout=subprocess.DEVNULL)
            assert filecmp.cmp("/tmp/mfx_vhb.u_a.out.vcf", "/tmp/mfx_vhb.u_b.out.vcf", shallow=False)
            assert filecmp.cmp("/tmp/mfx_vhb.u_a.log", "/tmp/mfx_vhb.u_b.log", shallow=False)
            assert os.path.getsize("/tmp/mfx_vhb.u_a.out.vcf") > 1000
    finally:
        for t in (".u_a", ".u_b", ".u_c"):
            for ext in (".out.vcf", ".log"):
                if os.path.exists("/tmp/mfx_vhb" + t + ext):
                    os.remove("/tmp/mfx_vhb" + t + ext)
        if os.path.exists("/tmp/mfx_vhb.vcf"):
            os.remove("/tmp/mfx_vhb.vcf")
