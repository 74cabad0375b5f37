import json,sys
d=sys.argv[1]
s=json.load(open(d+"/cornell_box.json"))
for o in s["objects"]:
    if o.get("material")=="white_plastic": o["material"]="white_wall"
json.dump(s,open(d+"/cornell_matte.json","w"))
s=json.load(open(d+"/cornell_box.json"))
for m in s["materials"]:
    if m["type"]=="matte": m.update({"type":"plastic","gloss":[0.6,0.6,0.6],"roughness":0.5})
json.dump(s,open(d+"/cornell_plastic.json","w"))
