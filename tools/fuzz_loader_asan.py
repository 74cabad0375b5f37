"""The scene loader (JSON, OBJ, MERL, animated_mesh keyframes, textures) built with AddressSanitizer + UndefinedBehaviorSanitizer
(tools/loader_check.cpp) over mutated scene files and mutated asset files:
    python tools/fuzz_loader_asan.py <seed> <n> <harness>"""
import copy, json, os, random, shutil, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tray_rust_amd import scenes
from fuzz_loader import mutate

seed, n, harness = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
rng = random.Random(seed)
d = tempfile.mkdtemp(prefix="fla")
base = os.path.join(d, "base"); os.makedirs(base)
scenes.write_assets(base, cornell=(32, 32, 4), small=(32, 32, 4))
scenes.write_moving_box(base, width=32, height=32, samples=4)
scenes.write_waving_flag(base, grid=5, n_keys=3, width=32, height=32, samples=4)
scenes.write_dragon_assets(base, film=(32, 32, 4), grid=6, extent=1.0)
scenes.write_textured_box(base, width=32, height=32, samples=4)
scene_files = [f for f in os.listdir(base) if f.endswith(".json")]
assets = []
for r, _, fs in os.walk(base):
    for f in fs:
        if not f.endswith(".json"):
            assets.append(os.path.relpath(os.path.join(r, f), base))
stats, fails, batch = {"ok": 0, "err": 0}, 0, []


def flush():
    global batch, fails
    if not batch:
        return
    r = subprocess.run([harness] + [os.path.join(b, s) for b, s in batch], capture_output=True, text=True, timeout=900)
    for line in r.stdout.splitlines():
        stats[line.split()[0]] = stats.get(line.split()[0], 0) + 1
    if r.returncode != 0:
        fails += 1
        for b, s in batch:
            rr = subprocess.run([harness, os.path.join(b, s)], capture_output=True, text=True, timeout=300)
            if rr.returncode != 0:
                print("FAIL", os.path.join(b, s), [l for l in rr.stderr.splitlines() if "runtime error" in l or "ERROR" in l][:2])
                break
    else:
        for b, _ in batch:
            shutil.rmtree(b, ignore_errors=True)
    batch = []


for i in range(n):
    sd = os.path.join(d, f"s{i}")
    shutil.copytree(base, sd, symlinks=True)
    which = rng.choice(scene_files)
    if i % 2 == 0:      # a structural mutation of the scene file
        doc = json.load(open(os.path.join(sd, which)))
        for _ in range(rng.randrange(1, 4)):
            mutate(doc, rng)
        json.dump(doc, open(os.path.join(sd, which), "w"))
    else:               # a byte-level mutation of one asset file
        victim = rng.choice(assets)
        b = bytearray(open(os.path.join(sd, victim), "rb").read())
        if b:
            m = rng.randrange(5)
            if m == 0: b = b[: rng.randrange(len(b) + 1)]
            elif m == 1:
                for _ in range(rng.randrange(1, 20)): b[rng.randrange(len(b))] = rng.randrange(256)
            elif m == 2 and victim.endswith(".obj"):
                lines = bytes(b).decode(errors="replace").split("\n"); k = rng.randrange(len(lines))
                lines[k] = rng.choice(["f 1 2 999999", "f -5 -6 -70000", "f 1/2/3 4//5 6/7", "v 1 2", "v a b c", "f 0 0 0", "f 1 2", "vn", "f 1 2 3 4 5", "o", "g x", "v 1e40 nan inf", "f 4294967296 1 2"])
                b = bytearray("\n".join(lines).encode())
            elif m == 3: b = b + bytes(rng.randrange(256) for _ in range(50))
            else:
                k = rng.randrange(max(1, len(b) - 12)); b[k:k + 12] = (2**31 - 1).to_bytes(4, "little") * 3
        open(os.path.join(sd, victim), "wb").write(bytes(b))
        which = {"models/flag": "waving_flag.json", "textures/": "textured_box.json", "brdfs/": "dragon.json", "models/dragon": "dragon.json"}.get(next((k for k in ("models/flag", "textures/", "brdfs/", "models/dragon") if victim.startswith(k)), ""), which)
    batch.append((sd, which))
    if len(batch) == 20:
        flush()
        if fails >= 3:
            break
flush()
print(stats, "failing batches:", fails)
