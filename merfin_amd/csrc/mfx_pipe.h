// mfx_pipe.h -- compressed file readers / writers chosen from the file name suffix, as meryl-utility's
// compressedFileReader / compressedFileWriter do (call sites: src/merfin/merfin.C:195, merfin-variants.C:149,
// merfin-histogram.C:151).  The compressor runs as a child started with posix_spawnp and an ARGV ARRAY: the path
// never passes through a shell, so quotes, spaces or `;` in a file name are just bytes of the name; and the
// child's exit status is reported by close() (a failed gzip is an I/O error, not a silent success).
// Header-only: shared by libmerfin_amd and the CLI's FASTA reader.
#pragma once
#include <fcntl.h>
#include <spawn.h>
#include <stdio.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

#include <string>

extern char **environ;

struct mfx_file {
  FILE *f = nullptr;
  pid_t pid = -1;            // > 0: a compressor child is attached
  bool  is_pipe() const { return pid > 0; }
};

inline const char *mfx_suffix_tool(const std::string &p) {
  auto ends = [&](const char *suf) { size_t n = strlen(suf); return p.size() >= n && p.compare(p.size() - n, n, suf) == 0; };
  return ends(".gz") ? "gzip" : ends(".bz2") ? "bzip2" : ends(".xz") ? "xz" : nullptr;
}

// read side: `tool -dc -- path` with its stdout on a pipe; plain files are fopen()ed
inline mfx_file mfx_open_reader(const char *path) {
  mfx_file r;
  const char *tool = mfx_suffix_tool(path);
  if (!tool) { r.f = fopen(path, "rb"); return r; }
  if (access(path, R_OK) != 0) return r;
  int fd[2];
  if (pipe2(fd, O_CLOEXEC) != 0) return r;      // dup2 onto 0/1 in the child clears the flag there only
  posix_spawn_file_actions_t fa;
  posix_spawn_file_actions_init(&fa);
  posix_spawn_file_actions_adddup2(&fa, fd[1], 1);
  char *argv[] = {(char *)tool, (char *)"-dc", (char *)"--", (char *)path, nullptr};
  pid_t pid = -1;
  const int rc = posix_spawnp(&pid, tool, &fa, nullptr, argv, environ);
  posix_spawn_file_actions_destroy(&fa);
  close(fd[1]);
  if (rc != 0) { close(fd[0]); return r; }
  r.f = fdopen(fd[0], "rb");
  r.pid = pid;
  if (!r.f) { close(fd[0]); waitpid(pid, nullptr, 0); r.pid = -1; }
  return r;
}

// write side: `tool -c` with stdin on a pipe and stdout on the (created / truncated / appended) file
inline mfx_file mfx_open_writer(const char *path, bool append) {
  mfx_file r;
  const char *tool = mfx_suffix_tool(path);
  if (!tool) { r.f = fopen(path, append ? "a" : "w"); return r; }
  const int out = open(path, O_WRONLY | O_CREAT | O_CLOEXEC | (append ? O_APPEND : O_TRUNC), 0666);
  if (out < 0) return r;
  int fd[2];
  if (pipe2(fd, O_CLOEXEC) != 0) { close(out); return r; }
  posix_spawn_file_actions_t fa;
  posix_spawn_file_actions_init(&fa);
  posix_spawn_file_actions_adddup2(&fa, fd[0], 0);
  posix_spawn_file_actions_adddup2(&fa, out, 1);
  char *argv[] = {(char *)tool, (char *)"-c", nullptr};
  pid_t pid = -1;
  const int rc = posix_spawnp(&pid, tool, &fa, nullptr, argv, environ);
  posix_spawn_file_actions_destroy(&fa);
  close(fd[0]);
  close(out);
  if (rc != 0) { close(fd[1]); return r; }
  r.f = fdopen(fd[1], "wb");
  r.pid = pid;
  if (!r.f) { close(fd[1]); waitpid(pid, nullptr, 0); r.pid = -1; }
  return r;
}

// 0 on success; non-zero when the stream had an error or the compressor did not exit cleanly.
// early_ok: the reader stopped before end of input (the child then dies of SIGPIPE, which is not an error)
inline int mfx_close(mfx_file &h, bool early_ok = false) {
  int bad = 0;
  if (h.f) {
    if (ferror(h.f)) bad = 1;
    if (fclose(h.f) != 0) bad = 1;
    h.f = nullptr;
  }
  if (h.pid > 0) {
    int st = 0;
    if (waitpid(h.pid, &st, 0) < 0) bad = 1;
    else if (WIFEXITED(st)) { if (WEXITSTATUS(st) != 0 && !early_ok) bad = 1; }
    else if (!(early_ok && WIFSIGNALED(st) && WTERMSIG(st) == SIGPIPE)) bad = 1;
    h.pid = -1;
  }
  return bad;
}
