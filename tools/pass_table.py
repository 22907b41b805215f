import csv, re, collections, sys
rows=list(csv.DictReader(open(sys.argv[1])))
def norm(n):
    n=n.replace("void ","").replace("mtsv::(anonymous namespace)::","")
    return re.sub(r"\(.*","",n)
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
passes=[];cur=None
for r in rows:
    k=norm(r["Kernel_Name"])
    k = "k_search" if k.startswith("k_search") else k
    if not k.startswith("k_") or k in ("k_expand_sa","k_kmer_level","k_kmer_level1"): continue
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
    if k=="k_search":
        cur={"grid":int(r["Grid_Size_X"]), "k":collections.OrderedDict(), "t0":int(r["Start_Timestamp"]), "t1":0}
        passes.append(cur)
    if cur is None: continue
    cur["k"][k]=cur["k"].get(k,0)+d
    cur["t1"]=int(r["End_Timestamp"])
n=int(sys.argv[2]) if len(sys.argv)>2 else 16
for p in passes[-n:]:
    reads=p["grid"]/18
    tot=sum(p["k"].values())
    print("reads~%8d span %.2f sum %.2f |"%(reads,(p["t1"]-p["t0"])/1e6,tot), " ".join(f"{k[2:12]}={v:.2f}" for k,v in p["k"].items()))
