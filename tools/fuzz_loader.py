"""Mutation fuzz of the scene loader: every mutated scene must load or fail with an error code -- never crash or hang."""
import copy, json, os, random, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def mutate(obj, rng, depth=0):
    """one random structural mutation somewhere in the tree"""
    paths = []
    def walk(o, path):
        paths.append(path)
        if isinstance(o, dict):
            for k in o: walk(o[k], path + [k])
        elif isinstance(o, list):
            for i in range(min(len(o), 6)): walk(o[i], path + [i])
    walk(obj, [])
    path = rng.choice(paths[1:])
    parent = obj
    for k in path[:-1]: parent = parent[k]
    key = path[-1]
    choice = rng.randrange(9)
    junk = [None, -1, 0, 1e30, -1e30, "", "x", [], {}, [1], [1, 2, 3, 4, 5], True, 2**40, 0.5, float("nan") if False else 1e-30, "matte", [[]], {"type": "x"}, 1e9, 65536, 7]
    if choice == 0 and isinstance(parent, dict): del parent[key]
    elif choice == 1 and isinstance(parent, list): del parent[key]
    elif choice == 2 and isinstance(parent, list): parent.append(copy.deepcopy(parent[key]))
    elif choice == 3 and isinstance(parent[key], list) and parent[key]: parent[key] = parent[key][: rng.randrange(len(parent[key]))]
    elif choice == 4 and isinstance(parent[key], (int, float)) and not isinstance(parent[key], bool): parent[key] = parent[key] * rng.choice([-1, 0, 1e6, 1e-6, 1000])
    else: parent[key] = copy.deepcopy(rng.choice(junk))
    return obj

def worker(files, out):
    import tray_rust_amd as T
    from tray_rust_amd import _lib as L
    res = []
    for f in files:
        try:
            scene, rt, spp, fi = T.Scene.load_file(f)
            for fr in (0, 3):
                try: scene.flatten(fr)
                except L.TrayError as e: pass
            res.append((f, "ok"))
        except L.TrayError as e:
            res.append((f, "err " + str(e)[:80]))
        except Exception as e:
            res.append((f, "pyexc " + repr(e)[:120]))
        with open(out, "w") as fh: json.dump(res, fh)

if __name__ == "__main__":
    if sys.argv[1] == "worker":
        worker(json.load(open(sys.argv[2])), sys.argv[3]); sys.exit(0)
    seed = int(sys.argv[1]); n = int(sys.argv[2]); base = sys.argv[3]
    rng = random.Random(seed)
    src = json.load(open(base))
    d = tempfile.mkdtemp(prefix="fz")
    # assets next to the scene
    for a in os.listdir(os.path.dirname(base)):
        if a.endswith((".obj", ".binary")): os.symlink(os.path.join(os.path.dirname(base), a), os.path.join(d, a))
    if os.path.isdir(os.path.join(os.path.dirname(base), "models")): os.symlink(os.path.join(os.path.dirname(base), "models"), os.path.join(d, "models"))
    files = []
    for i in range(n):
        o = copy.deepcopy(src)
        for _ in range(rng.choice([1, 1, 1, 2, 3])): 
            try: mutate(o, rng)
            except Exception: pass
        f = os.path.join(d, f"m{i}.json"); json.dump(o, open(f, "w")); files.append(f)
    # run in batches in subprocesses so a crash is attributable
    todo = files; bad = []; stats = {}
    while todo:
        lst = os.path.join(d, "list.json"); out = os.path.join(d, "out.json")
        json.dump(todo, open(lst, "w"))
        if os.path.exists(out): os.remove(out)
        try:
            p = subprocess.run([sys.executable, __file__, "worker", lst, out], capture_output=True, timeout=120 + 2 * len(todo))
            rc = p.returncode
        except subprocess.TimeoutExpired:
            rc = "timeout"
        done = json.load(open(out)) if os.path.exists(out) else []
        for f, r in done: stats[r.split(" ")[0]] = stats.get(r.split(" ")[0], 0) + 1
        for f, r in done:
            if r.startswith("pyexc"): bad.append((f, r))
        if rc == 0: break
        culprit = todo[len(done)] if len(done) < len(todo) else None
        bad.append((culprit, f"CRASH rc={rc}"))
        todo = todo[len(done) + 1:]
    print(stats)
    for b in bad: print(b)
