// faultwhere.c — where do a thread's minor page faults land?  (two batches of 16 faults per frame on the caller's thread)
// A perf software event (PERF_COUNT_SW_PAGE_FAULTS_MIN, period 1) on the calling thread samples every fault's instruction pointer
// and data address; fw_stop() maps the addresses onto /proc/self/maps and the ips onto symbols (dladdr) and prints histograms.
// build: gcc -O2 -shared -fPIC -o libfaultwhere.so faultwhere.c -ldl        use: bench.py with PLV_BENCH_FAULTWHERE=<path to the .so>
#define _GNU_SOURCE
#include <dlfcn.h>
#include <errno.h>
#include <linux/perf_event.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/ioctl.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#define FW_PAGES 512  // ring: 2 MB = ~50 000 samples of 40 bytes

static int g_fd = -1;
static void *g_ring = NULL;
static size_t g_ring_bytes = 0;
static uint64_t g_marks[4096];
static int g_nmarks = 0;

static uint64_t now_ns(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (uint64_t)ts.tv_sec * 1000000000ull + ts.tv_nsec;
}

int fw_start(void) {
  struct perf_event_attr at;
  memset(&at, 0, sizeof at);
  at.size = sizeof at;
  at.type = PERF_TYPE_SOFTWARE;
  at.config = PERF_COUNT_SW_PAGE_FAULTS_MIN;
  at.sample_period = 1;
  at.sample_type = PERF_SAMPLE_IP | PERF_SAMPLE_TID | PERF_SAMPLE_TIME | PERF_SAMPLE_ADDR;
  at.disabled = 1;
  at.use_clockid = 1;
  at.clockid = CLOCK_MONOTONIC;
  at.exclude_hv = 1;
  int fd = (int)syscall(SYS_perf_event_open, &at, 0, -1, -1, 0);
  if (fd < 0) {
    at.exclude_kernel = 1;  // (faults taken inside system calls are not seen then)
    fd = (int)syscall(SYS_perf_event_open, &at, 0, -1, -1, 0);
    fprintf(stderr, "[faultwhere] kernel-side faults excluded\n");
  }
  if (fd < 0) {
    fprintf(stderr, "[faultwhere] perf_event_open: %s\n", strerror(errno));
    return -1;
  }
  g_ring_bytes = (size_t)(1 + FW_PAGES) * 4096;
  g_ring = mmap(NULL, g_ring_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  if (g_ring == MAP_FAILED) {
    fprintf(stderr, "[faultwhere] mmap of the ring: %s\n", strerror(errno));
    close(fd);
    g_ring = NULL;
    return -1;
  }
  // touch the ring now: its own first-touch faults would land in the sample otherwise
  for (size_t i = 0; i < g_ring_bytes; i += 4096) (void)((volatile char *)g_ring)[i];
  g_fd = fd;
  g_nmarks = 0;
  ioctl(fd, PERF_EVENT_IOC_RESET, 0);
  ioctl(fd, PERF_EVENT_IOC_ENABLE, 0);
  return 0;
}

void fw_mark(void) {  // a step starts (the odd system call is the mark tools/ubench/systrace.c looks for)
  if (g_nmarks < 4096) g_marks[g_nmarks++] = now_ns();
  syscall(SYS_getpgid, 424242);
}

void fw_end(void) { syscall(SYS_getpgid, 424243); }  // a step has ended

struct map_ent {
  uint64_t a, b;
  char perm[8];
  char name[200];
  int hits;
};

struct sample {
  uint64_t ip, t, addr;
  uint32_t tid;
};

int fw_stop(const char *out_path) {
  if (g_fd < 0) return -1;
  ioctl(g_fd, PERF_EVENT_IOC_DISABLE, 0);
  uint64_t t_end = now_ns();
  FILE *out = out_path && out_path[0] ? fopen(out_path, "w") : stderr;
  if (!out) out = stderr;
  // the maps
  static struct map_ent maps[8192];
  int nm = 0;
  FILE *mf = fopen("/proc/self/maps", "r");
  char line[512];
  while (mf && fgets(line, sizeof line, mf) && nm < 8192) {
    unsigned long a, b;
    char perm[8] = {0}, name[200] = {0};
    if (sscanf(line, "%lx-%lx %7s %*s %*s %*s %199[^\n]", &a, &b, perm, name) >= 3) {
      maps[nm].a = a, maps[nm].b = b, maps[nm].hits = 0;
      strcpy(maps[nm].perm, perm);
      strcpy(maps[nm].name, name[0] ? name : "[anon]");
      ++nm;
    }
  }
  if (mf) fclose(mf);
  struct perf_event_mmap_page *hdr = (struct perf_event_mmap_page *)g_ring;
  const char *data = (const char *)g_ring + 4096;
  const uint64_t dsz = (uint64_t)FW_PAGES * 4096;
  uint64_t head = hdr->data_head, tail = hdr->data_tail;
  __sync_synchronize();
  static struct sample smp[65536];
  int ns = 0, lost = 0;
  while (tail < head && ns < 65536) {
    struct perf_event_header h;
    char rec[256];
    for (size_t i = 0; i < sizeof h; ++i) ((char *)&h)[i] = data[(tail + i) % dsz];
    if (h.size == 0) break;
    size_t n = h.size < sizeof rec ? h.size : sizeof rec;
    for (size_t i = 0; i < n; ++i) rec[i] = data[(tail + i) % dsz];
    if (h.type == PERF_RECORD_SAMPLE) {
      const char *p = rec + sizeof h;
      memcpy(&smp[ns].ip, p, 8);
      memcpy(&smp[ns].tid, p + 12, 4);
      memcpy(&smp[ns].t, p + 16, 8);
      memcpy(&smp[ns].addr, p + 24, 8);
      ++ns;
    } else if (h.type == PERF_RECORD_LOST) {
      ++lost;
    }
    tail += h.size;
  }
  fprintf(out, "[faultwhere] %d faults sampled over %d marked steps (%d lost records)\n", ns, g_nmarks, lost);
  // per step: offset of each fault from the step's start, the mapping of its address, the symbol of its ip
  int shown = 0;
  for (int s = 0; s < ns; ++s) {
    int step = -1;
    for (int m = 0; m < g_nmarks; ++m)
      if (g_marks[m] <= smp[s].t) step = m;
    const struct map_ent *me = NULL;
    for (int m = 0; m < nm; ++m)
      if (smp[s].addr >= maps[m].a && smp[s].addr < maps[m].b) {
        maps[m].hits++;
        me = &maps[m];
        break;
      }
    const int last_steps = step >= g_nmarks - 4;
    if (last_steps && shown < 400) {
      Dl_info di;
      memset(&di, 0, sizeof di);
      const int okd = dladdr((void *)smp[s].ip, &di);
      const char *lib = okd && di.dli_fname ? strrchr(di.dli_fname, '/') : NULL;
      fprintf(out, "step %4d +%8.1f us  addr %012lx  in %-40s %s (%ld pages, +%ld)  ip %012lx %s+0x%lx %s\n", step,
              step >= 0 ? (smp[s].t - g_marks[step]) / 1e3 : 0.0, (unsigned long)smp[s].addr, me ? me->name : "?", me ? me->perm : "",
              me ? (long)((me->b - me->a) / 4096) : 0L, me ? (long)((smp[s].addr - me->a) / 4096) : 0L, (unsigned long)smp[s].ip,
              lib ? lib + 1 : "?", okd ? (unsigned long)(smp[s].ip - (uint64_t)di.dli_fbase) : 0ul, okd && di.dli_sname ? di.dli_sname : "");
      ++shown;
    }
  }
  fprintf(out, "[faultwhere] by mapping (whole run, %.1f ms):\n", g_nmarks ? (t_end - g_marks[0]) / 1e6 : 0.0);
  for (int rep = 0; rep < 25; ++rep) {
    int best = -1;
    for (int m = 0; m < nm; ++m)
      if (maps[m].hits > 0 && (best < 0 || maps[m].hits > maps[best].hits)) best = m;
    if (best < 0) break;
    fprintf(out, "  %6d faults  %012lx-%012lx %s %6ld pages  %s\n", maps[best].hits, (unsigned long)maps[best].a, (unsigned long)maps[best].b,
            maps[best].perm, (long)((maps[best].b - maps[best].a) / 4096), maps[best].name);
    maps[best].hits = 0;
  }
  if (out != stderr) fclose(out);
  munmap(g_ring, g_ring_bytes);
  close(g_fd);
  g_fd = -1;
  g_ring = NULL;
  return ns;
}
