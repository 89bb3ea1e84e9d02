import sys, runpy, atexit, collections
def dump():
    sizes = collections.Counter()
    lines = open("/proc/self/maps").read().splitlines()
    for l in lines:
        parts = l.split()
        a, b = (int(x, 16) for x in parts[0].split("-"))
        name = parts[5] if len(parts) > 5 else "[anon]"
        if "dri" in name or "kfd" in name or (b - a) == 176 * 4096 or (b - a) == 16 * 4096:
            sizes[(name, (b - a) // 4096, parts[1])] += 1
    for k, v in sorted(sizes.items(), key=lambda kv: -kv[1])[:40]:
        print("[maps]", v, "x", k, file=sys.stderr)
atexit.register(dump)
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
