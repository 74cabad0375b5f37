"""Mutated texture files (PNG, baseline / progressive JPEG, GIF, PPM, BMP, TGA, TIFF, ICO, Radiance HDR, WebP) through the loader's own decoders (csrc/host/image.hpp): every
file must come back as a picture or as an error code -- no crash, no hang, no allocation from a forged header.
    python tools/fuzz_images.py <seed> <n>           needs Pillow to write the valid originals"""
import json, os, random, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(d, n, out):
    import tray_rust_amd as T
    from tray_rust_amd import _lib as L
    res = []
    for i in range(n):
        f = os.path.join(d, f"s{i}", "textured_box.json")
        try:
            scene, *_ = T.Scene.load_file(f); scene.flatten(0); res.append("ok")
        except L.TrayError:
            res.append("err")
        except Exception as e:
            res.append("pyexc " + repr(e)[:100])
        json.dump(res, open(out, "w"))


if sys.argv[1] == "worker":
    worker(sys.argv[2], int(sys.argv[3]), sys.argv[4]); sys.exit(0)
import numpy as np
from PIL import Image
from tray_rust_amd import scenes
rng = random.Random(int(sys.argv[1])); n = int(sys.argv[2])
d = tempfile.mkdtemp(prefix="fi")
base = os.path.join(d, "base")
path = scenes.write_textured_box(base, width=64, height=64, samples=4)
doc = json.load(open(path))
tex = [t for t in doc["textures"] if t.get("type") == "image"][0]      # the file this texture names is replaced by the victims below
yy, xx = np.mgrid[0:40, 0:56]
pix = np.stack([128 + 100 * np.sin(xx / 5.0) * np.cos(yy / 7.0), 128 + 90 * np.cos(xx / 3.0 + yy / 11.0), 40 + 3 * xx + 2 * yy], axis=2).clip(0, 255).astype(np.uint8)
originals = {}
tmp = os.path.join(d, "orig"); os.makedirs(tmp)
Image.fromarray(pix, "RGB").save(os.path.join(tmp, "a.jpg"), quality=85, subsampling=2); originals["a.jpg"] = None
Image.fromarray(pix, "RGB").save(os.path.join(tmp, "p.jpg"), quality=85, subsampling=2, progressive=True); originals["p.jpg"] = None
Image.fromarray(pix, "RGB").save(os.path.join(tmp, "r.jpg"), quality=60, subsampling=0, progressive=True, restart_marker_blocks=2); originals["r.jpg"] = None
Image.fromarray(pix, "RGB").save(os.path.join(tmp, "c.png")); originals["c.png"] = None
Image.fromarray(pix, "RGB").quantize(64).save(os.path.join(tmp, "g.gif")); originals["g.gif"] = None
Image.fromarray(pix, "RGB").quantize(200).save(os.path.join(tmp, "i.gif"), interlace=1, transparency=5); originals["i.gif"] = None
Image.fromarray(pix, "RGB").save(os.path.join(tmp, "b.bmp")); originals["b.bmp"] = None
Image.fromarray(pix, "RGB").save(os.path.join(tmp, "t.tga")); originals["t.tga"] = None
Image.fromarray(pix, "RGB").save(os.path.join(tmp, "z.tif"), compression="tiff_lzw"); originals["z.tif"] = None
Image.fromarray(pix, "RGB").save(os.path.join(tmp, "k.tif"), compression="packbits"); originals["k.tif"] = None
Image.fromarray(pix, "RGB").convert("RGBA").resize((32, 32)).save(os.path.join(tmp, "o.ico"), sizes=[(16, 16), (32, 32)], bitmap_format="bmp"); originals["o.ico"] = None
Image.fromarray(pix, "RGB").save(os.path.join(tmp, "x.webp"), quality=90, method=4); originals["x.webp"] = None
open(os.path.join(tmp, "h.hdr"), "wb").write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (pix.shape[0], pix.shape[1]) + np.concatenate([pix, np.full(pix.shape[:2] + (1,), 129, np.uint8)], axis=2).tobytes()); originals["h.hdr"] = None
for k in originals:
    originals[k] = open(os.path.join(tmp, k), "rb").read()
names = sorted(originals)
for i in range(n):
    sd = os.path.join(d, f"s{i}")
    subprocess.run(["cp", "-r", base, sd], check=True)
    victim = names[i % len(names)]
    b = bytearray(originals[victim]); m = rng.randrange(6)
    if m == 0:
        b = b[: rng.randrange(len(b) + 1)]
    elif m == 1:
        for _ in range(rng.randrange(1, 12)):
            b[rng.randrange(len(b))] = rng.randrange(256)
    elif m == 2:      # in the headers, where sizes and table definitions live
        for _ in range(rng.randrange(1, 6)):
            b[rng.randrange(min(len(b), 700))] = rng.choice([0, 1, 0x7f, 0x80, 0xff, rng.randrange(256)])
    elif m == 3:
        k = rng.randrange(max(1, len(b) - 8)); b[k:k + 4] = b"\xff\xff\xff\x7f"
    elif m == 4:
        k = rng.randrange(len(b)); b[k:k] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 40)))
    else:
        k = rng.randrange(len(b)); del b[k:k + rng.randrange(1, 60)]
    dd = json.loads(json.dumps(doc))
    for t in dd["textures"]:
        if t.get("name") == tex["name"]:
            t["file"] = "textures/victim" + os.path.splitext(victim)[1]
    open(os.path.join(sd, "textures", "victim" + os.path.splitext(victim)[1]), "wb").write(bytes(b))
    json.dump(dd, open(os.path.join(sd, "textured_box.json"), "w"))
out = os.path.join(d, "out.json"); stats = {}
p = subprocess.run([sys.executable, __file__, "worker", d, str(n), out], capture_output=True, timeout=1200)
res = json.load(open(out)) if os.path.exists(out) else []
for r in res:
    stats[r.split(" ")[0]] = stats.get(r.split(" ")[0], 0) + 1
print(stats, "rc", p.returncode, "done", len(res), "of", n, p.stderr.decode()[-300:] if p.returncode else "")
print([r for r in res if r.startswith("pyexc")][:5])
if p.returncode and len(res) < n:
    print("CRASH at", os.path.join(d, f"s{len(res)}"))
