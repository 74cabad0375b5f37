import json, os, random, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def worker(d, n, out):
    import tray_rust_amd as T
    from tray_rust_amd import _lib as L
    res = []
    for i in range(n):
        f = os.path.join(d, f"s{i}", "dragon.json")
        try:
            scene, *_ = T.Scene.load_file(f); scene.flatten(0); res.append("ok")
        except L.TrayError as e: res.append("err")
        except Exception as e: res.append("pyexc " + repr(e)[:100])
        json.dump(res, open(out, "w"))
if sys.argv[1] == "worker":
    worker(sys.argv[2], int(sys.argv[3]), sys.argv[4]); sys.exit(0)
from tray_rust_amd import scenes
rng = random.Random(int(sys.argv[1])); n = int(sys.argv[2])
d = tempfile.mkdtemp(prefix="fa")
base = os.path.join(d, "base"); os.makedirs(base)
scenes.write_dragon_assets(base, film=(64, 64, 4), grid=6, extent=1.0)
names = []
for r, _, fs in os.walk(base):
    for f in fs: names.append(os.path.relpath(os.path.join(r, f), base))
print(names)
for i in range(n):
    sd = os.path.join(d, f"s{i}"); os.makedirs(sd)
    victim = rng.choice([x for x in names if not x.endswith(".json")])
    for x in names:
        data = open(os.path.join(base, x), "rb").read()
        if x == victim:
            b = bytearray(data); m = rng.randrange(6)
            if m == 0: b = b[: rng.randrange(len(b) + 1)]
            elif m == 1:
                for _ in range(rng.randrange(1, 20)): b[rng.randrange(len(b))] = rng.randrange(256)
            elif m == 2 and x.endswith(".obj"):
                lines = data.decode().split("\n"); k = rng.randrange(len(lines))
                lines[k] = rng.choice(["f 1 2 999999", "f -5 -6 -70000", "f 1/2/3 4//5 6/7", "v 1 2", "v a b c", "f 0 0 0", "f 1 2", "vn", "f 1 2 3 4 5", "o", "g x", "v 1e40 nan inf", "f 4294967296 1 2"])
                b = bytearray("\n".join(lines).encode())
            elif m == 3: b = bytearray()
            elif m == 4: b = b + bytes(rng.randrange(256) for _ in range(50))
            else:
                k = rng.randrange(max(1, len(b) - 12)); b[k:k+12] = (2**31 - 1).to_bytes(4, "little") * 3
            data = bytes(b)
        os.makedirs(os.path.dirname(os.path.join(sd, x)), exist_ok=True)
        open(os.path.join(sd, x), "wb").write(data)
out = os.path.join(d, "out.json"); start = 0; stats = {}; bad = []
p = subprocess.run([sys.executable, __file__, "worker", d, str(n), out], capture_output=True, timeout=600)
res = json.load(open(out)) if os.path.exists(out) else []
for r in res: stats[r.split(" ")[0]] = stats.get(r.split(" ")[0], 0) + 1
print(stats, "rc", p.returncode, "done", len(res), "of", n, p.stderr.decode()[-300:] if p.returncode else "")
print([r for r in res if r.startswith("pyexc")][:5])
