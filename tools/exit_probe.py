import subprocess, time, sys
exe = "/root/repo/merfin_amd/bin/merfin"
for env in ({}, {"MFX_CLI_QUICK_EXIT": "1"}):
    for i in range(3):
        t0 = time.time()
        r = subprocess.run([exe, "-hist", "-sequence", "/nonexistent.fa", "-readmers", "/nonexistent.mfxk", "-peak", "26", "-output", "/tmp/x.hist"], capture_output=True, text=True)
        print("rc %d wall %.3f s  %s" % (r.returncode, time.time() - t0, r.stderr.strip().splitlines()[-1][:80] if r.stderr.strip() else ""), flush=True)
import ctypes
t0 = time.time()
r = subprocess.run([sys.executable, "-c", "import ctypes,time; h=ctypes.CDLL('/opt/rocm/lib/libamdhip64.so'); n=ctypes.c_int(0); t=time.time(); h.hipGetDeviceCount(ctypes.byref(n)); print('hipGetDeviceCount %.3f s' % (time.time()-t), n.value, flush=True); t=time.time(); import os; os._exit(0)"], capture_output=True, text=True)
print("python hip-only child: wall %.3f s; %s" % (time.time() - t0, r.stdout.strip()))
