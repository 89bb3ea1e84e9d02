import ctypes, os, sys, runpy
libc = ctypes.CDLL("libc.so.6", use_errno=True)
MCL_CURRENT, MCL_FUTURE = 1, 2
rc = libc.mlockall(MCL_CURRENT | MCL_FUTURE)
print("mlockall rc", rc, ctypes.get_errno(), file=sys.stderr)
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
