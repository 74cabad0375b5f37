"""Mutated PNG / JPEG (baseline, progressive) / GIF / BMP / TGA / TIFF / ICO / Radiance HDR / WebP files through tools/image_decode_check (image.hpp built with
-fsanitize=address,undefined): any memory error or undefined arithmetic aborts the harness and is reported with the file that caused it.
    g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all tools/image_decode_check.cpp -o /tmp/image_decode_check
    python tools/fuzz_images_asan.py <seed> <n> [harness]"""
import os, random, subprocess, tempfile, sys
import numpy as np
from PIL import Image
seed=int(sys.argv[1]); N=int(sys.argv[2]); HARNESS=sys.argv[3] if len(sys.argv) > 3 else "/tmp/image_decode_check"
d=tempfile.mkdtemp(prefix="fia")
yy, xx = np.mgrid[0:40, 0:56]
pix = np.stack([128 + 100 * np.sin(xx / 5.0) * np.cos(yy / 7.0), 128 + 90 * np.cos(xx / 3.0 + yy / 11.0), 40 + 3 * xx + 2 * yy], axis=2).clip(0, 255).astype(np.uint8)
im=Image.fromarray(pix,"RGB")
files={}
def save(name, **kw):
    p=os.path.join(d,name); (kw.pop('img',im)).save(p, **kw); files[name]=open(p,'rb').read()
save("a.jpg", quality=85, subsampling=2); save("p.jpg", quality=85, subsampling=2, progressive=True)
save("r.jpg", quality=60, subsampling=0, progressive=True, restart_marker_blocks=2); save("q.jpg", quality=30, subsampling=1, progressive=True)
save("c.png"); save("g.gif", img=im.quantize(64)); save("i.gif", img=im.quantize(200), interlace=1, transparency=5)
save("b.bmp"); save("t.tga"); save("l.jpg", img=Image.fromarray(pix[...,0],"L"), progressive=True)
save("u.tif", compression="raw"); save("z.tif", compression="tiff_lzw"); save("k.tif", compression="packbits"); save("d.tif", compression="tiff_lzw", tiffinfo={317: 2})
save("m.tif", img=im.quantize(40), compression="tiff_lzw"); save("w.tif", img=Image.fromarray((pix[...,0].astype(np.uint16)*257)), compression="tiff_lzw")
save("x.webp", quality=100, method=6); save("y.webp", quality=40, method=2); save("v.webp", img=Image.fromarray(np.random.default_rng(2).integers(0, 256, (40, 56, 3), dtype=np.uint8), "RGB"), quality=80, method=4)
save("n.ico", img=im.convert("RGBA").resize((32,32)), sizes=[(16,16),(32,32)]); save("o.ico", img=im.convert("RGBA").resize((32,32)), sizes=[(16,16),(32,32)], bitmap_format="bmp")
def hdr_bytes():
    w,h=pix.shape[1],pix.shape[0]; out=b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n"%(h,w)
    for y in range(h):
        out+=bytes([2,2,w>>8,w&255])
        for ch in range(4):
            row=(pix[y,:,ch] if ch<3 else np.full(w,129,np.uint8)); x=0
            while x<w:
                n=min(9,w-x); out+=(bytes([128+n,int(row[x])]) if (x//9)%2 else bytes([n])+row[x:x+n].tobytes()); x+=n
    return out
files["h.hdr"]=hdr_bytes()
rng=random.Random(seed)
names=sorted(files)
stats={'ok':0,'err':0}
batch=[]; fails=0
for i in range(N):
    v=names[i%len(names)]; b=bytearray(files[v]); m=rng.randrange(6)
    if m==0: b=b[:rng.randrange(len(b)+1)]
    elif m==1:
        for _ in range(rng.randrange(1,12)): b[rng.randrange(len(b))]=rng.randrange(256)
    elif m==2:
        for _ in range(rng.randrange(1,6)): b[rng.randrange(min(len(b),700))]=rng.choice([0,1,0x7f,0x80,0xff,rng.randrange(256)])
    elif m==3:
        k=rng.randrange(max(1,len(b)-8)); b[k:k+4]=b"\xff\xff\xff\x7f"
    elif m==4:
        k=rng.randrange(len(b)); b[k:k]=bytes(rng.randrange(256) for _ in range(rng.randrange(1,40)))
    else:
        k=rng.randrange(len(b)); del b[k:k+rng.randrange(1,60)]
    p=os.path.join(d,"m%05d"%i+os.path.splitext(v)[1]); open(p,'wb').write(bytes(b)); batch.append(p)
    if len(batch)==50 or i==N-1:
        r=subprocess.run([HARNESS]+batch,capture_output=True,text=True,errors='replace',timeout=600)
        for line in r.stdout.splitlines(): stats[line.split()[0]]+=1
        if r.returncode!=0:
            fails+=1
            for q in batch:
                rr=subprocess.run([HARNESS,q],capture_output=True,text=True,errors='replace',timeout=60)
                if rr.returncode!=0:
                    print("FAIL",q, [l for l in rr.stderr.splitlines() if 'runtime error' in l or 'ERROR' in l][:2]); break
            if fails>=3: break
        else:
            for q in batch: os.remove(q)
        batch=[]
print(stats,"failing batches:",fails)
