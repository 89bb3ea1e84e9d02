"""measurement aid: runs a script after glibc's malloc has been told never to trim or mmap (mallopt), to see what page faults remain"""
import ctypes, sys, runpy
libc = ctypes.CDLL("libc.so.6")
M_TRIM_THRESHOLD, M_TOP_PAD, M_MMAP_THRESHOLD, M_ARENA_MAX = -1, -2, -3, -8
print("mallopt", libc.mallopt(M_TRIM_THRESHOLD, 1 << 30), libc.mallopt(M_TOP_PAD, 256 << 20), libc.mallopt(M_MMAP_THRESHOLD, 1 << 30), file=sys.stderr)
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
