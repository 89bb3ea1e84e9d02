// systrace.c — a minimal system-call tracer (no strace in the image): which system calls does the thread that runs the frame make
// inside a step?  (system time and kernel-side page faults of the caller's thread.)
// The traced program marks the start of a step with getpgid(424242) (tools/ubench/faultwhere.c: fw_mark); the tracer lists, for the
// LAST steps, every system call of the marking thread between two marks: offset, number, first arguments, result, time inside.
// build: gcc -O2 -o systrace systrace.c      use: systrace <out.txt> <steps to list> -- python3 bench.py ...
// The child is started with PTRACE_TRACEME before it has touched the GPU; threads and children are followed.
#define _GNU_SOURCE
#include <errno.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/ptrace.h>
#include <sys/syscall.h>
#include <sys/types.h>
#include <sys/user.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#define MARK_ARG 424242
#define MAXREC 400000

struct rec {
  uint64_t t_in, t_out;
  long nr, a0, a1, a2, ret;
  int mark;
};
static struct rec *recs;
static int nrec = 0;

static uint64_t now_ns(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (uint64_t)ts.tv_sec * 1000000000ull + ts.tv_nsec;
}

static const char *name_of(long nr) {
  switch (nr) {
    case SYS_read: return "read";
    case SYS_write: return "write";
    case SYS_mmap: return "mmap";
    case SYS_mprotect: return "mprotect";
    case SYS_munmap: return "munmap";
    case SYS_brk: return "brk";
    case SYS_ioctl: return "ioctl";
    case SYS_madvise: return "madvise";
    case SYS_futex: return "futex";
    case SYS_sched_yield: return "sched_yield";
    case SYS_getrusage: return "getrusage";
    case SYS_clock_gettime: return "clock_gettime";
    case SYS_getpgid: return "getpgid";
    case SYS_mremap: return "mremap";
    case SYS_poll: return "poll";
    case SYS_nanosleep: return "nanosleep";
    case SYS_clock_nanosleep: return "clock_nanosleep";
    case SYS_rt_sigprocmask: return "rt_sigprocmask";
    case SYS_mlock: return "mlock";
    case SYS_munlock: return "munlock";
    case SYS_sched_getaffinity: return "sched_getaffinity";
    case SYS_membarrier: return "membarrier";
    default: return "?";
  }
}

int main(int argc, char **argv) {
  if (argc < 5 || strcmp(argv[3], "--")) {
    fprintf(stderr, "usage: systrace <out.txt> <steps to list> -- program args...\n");
    return 2;
  }
  const char *out_path = argv[1];
  const int list_steps = atoi(argv[2]);
  pid_t child = fork();
  if (child == 0) {
    ptrace(PTRACE_TRACEME, 0, 0, 0);
    raise(SIGSTOP);
    execvp(argv[4], argv + 4);
    perror("execvp");
    _exit(127);
  }
  recs = (struct rec *)calloc(MAXREC, sizeof *recs);
  int status;
  waitpid(child, &status, 0);
  ptrace(PTRACE_SETOPTIONS, child, 0, PTRACE_O_TRACESYSGOOD | PTRACE_O_TRACECLONE | PTRACE_O_TRACEFORK | PTRACE_O_TRACEVFORK | PTRACE_O_TRACEEXEC);
  ptrace(PTRACE_SYSCALL, child, 0, 0);
  pid_t marker = 0;   // the thread that marks the steps
  int in_call = 0;    // (of the marker thread) between entry and exit
  int exit_code = 0;
  // entry/exit state of every other thread is not needed: only the marker thread's calls are decoded
  static unsigned char inside[1 << 22];  // by tid (pid_max <= 4194304)
  for (;;) {
    pid_t p = waitpid(-1, &status, __WALL);
    if (p < 0) {
      if (errno == ECHILD) break;
      continue;
    }
    if (WIFEXITED(status) || WIFSIGNALED(status)) {
      if (p == child) exit_code = WIFEXITED(status) ? WEXITSTATUS(status) : 128 + WTERMSIG(status);
      continue;
    }
    if (!WIFSTOPPED(status)) continue;
    const int sig = WSTOPSIG(status);
    long deliver = 0;
    if (sig == (SIGTRAP | 0x80)) {
      unsigned char *st = &inside[p & ((1 << 22) - 1)];
      const int entering = !*st;
      *st = (unsigned char)entering;
      if (marker == 0 || p == marker) {
        struct user_regs_struct r;
        if (ptrace(PTRACE_GETREGS, p, 0, &r) == 0) {
          if (entering) {
            if (marker == 0 && (long)r.orig_rax == SYS_getpgid && (long)r.rdi == MARK_ARG) marker = p;
            if (p == marker && nrec < MAXREC) {
              struct rec *c = &recs[nrec];
              c->t_in = now_ns();
              c->nr = (long)r.orig_rax, c->a0 = (long)r.rdi, c->a1 = (long)r.rsi, c->a2 = (long)r.rdx;
              c->mark = c->nr == SYS_getpgid ? (c->a0 == MARK_ARG ? 1 : c->a0 == MARK_ARG + 1 ? 2 : 0) : 0;
              in_call = 1;
            }
          } else if (p == marker && in_call && nrec < MAXREC) {
            recs[nrec].t_out = now_ns();
            recs[nrec].ret = (long)r.rax;
            ++nrec;
            in_call = 0;
          }
        }
      }
    } else if (sig == SIGTRAP && (status >> 16) != 0) {
      // clone / fork / exec event: the new thread is attached automatically and starts stopped
    } else if (sig == SIGSTOP && p != child) {
      // the initial stop of an auto-attached thread
    } else if (sig != SIGTRAP) {
      deliver = sig;
    }
    ptrace(PTRACE_SYSCALL, p, 0, deliver);
  }
  FILE *out = fopen(out_path, "w");
  if (!out) out = stderr;
  int nmarks = 0;
  for (int i = 0; i < nrec; ++i) nmarks += recs[i].mark == 1;
  fprintf(out, "[systrace] %d system calls of the marking thread recorded, %d steps marked\n", nrec, nmarks);
  // histogram over all steps (calls between a mark and the next one), and the listing of the last steps
  long cnt[512] = {0};
  double tin[512] = {0};
  int seen = 0, inside_step = 0;
  uint64_t t_mark = 0;
  for (int i = 0; i < nrec; ++i) {
    if (recs[i].mark == 1) {
      ++seen;
      inside_step = 1;
      t_mark = recs[i].t_out;
      if (seen > nmarks - list_steps) fprintf(out, "---- step %d\n", seen - 1);
      continue;
    }
    if (recs[i].mark == 2) {
      if (seen > nmarks - list_steps) fprintf(out, "+%9.1f us  end of the step\n", (recs[i].t_in - t_mark) / 1e3);
      inside_step = 0;
      continue;
    }
    if (!seen || !inside_step) continue;
    if (recs[i].nr >= 0 && recs[i].nr < 512) {
      cnt[recs[i].nr]++;
      tin[recs[i].nr] += (recs[i].t_out - recs[i].t_in) / 1e3;
    }
    if (seen > nmarks - list_steps)
      fprintf(out, "+%9.1f us  %-16s(%3ld)  a0 %lx  a1 %lx  a2 %lx  = %ld   [%.1f us under the tracer]\n", (recs[i].t_in - t_mark) / 1e3, name_of(recs[i].nr),
              recs[i].nr, recs[i].a0, recs[i].a1, recs[i].a2, recs[i].ret, (recs[i].t_out - recs[i].t_in) / 1e3);
  }
  fprintf(out, "[systrace] per step, all steps:\n");
  for (int n = 0; n < 512; ++n)
    if (cnt[n]) fprintf(out, "  %-16s(%3d)  %.1f calls per step\n", name_of(n), n, (double)cnt[n] / (nmarks > 0 ? nmarks : 1));
  if (out != stderr) fclose(out);
  return exit_code;
}
