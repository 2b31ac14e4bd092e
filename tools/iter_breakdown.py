import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
o, v = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (20, 200)
D = {"o": o, "v": v}
shape = {"t1": "ov", "t2": "oovv", "c": "oovv", "asym": "oovv", "w_oovv": "oovv", "v_oovv": "oovv", "v_ovov": "ovov",
         "v_vvov": "vvov", "w_vvov": "vvov", "v_oovo": "oovo", "w_oovo": "oovo", "v_oooo": "oooo", "v_vvvv": "vvvv",
         "I_vo": "vo", "I_vv": "vv", "I_oo_p": "oo", "I_oo": "oo", "I_oooo": "oooo", "I_ovov": "ovov", "I_voov": "voov",
         "x_voov": "voov", "I_vovv_p": "vovv", "I_ooov_p": "ooov", "r1": "ov", "r2": "oovv"}
sites = [
 ("w_oovv","miea","t1","me","I_vo","ai"), ("w_vvov","ebma","t1","me","I_vv","ba"), ("w_oovv","mneb","c","mnea","I_vv","ba"),
 ("w_oovo","miej","t1","me","I_oo_p","ji"), ("asym","mjef","v_oovv","mief","I_oo_p","ji"), ("t1","je","I_vo","ei","I_oo","ji"),
 ("c","klef","v_oovv","ijef","I_oooo","klij"), ("t1","ke","v_oovo","ilej","I_oooo","klij"), ("t1","le","v_oovo","jkei","I_oooo","klij"),
 ("v_oovv","mibe","c","mjae","I_ovov","jbia"), ("v_oovo","mibj","t1","ma","I_ovov","jbia"), ("t1","je","v_vvov","ebia","I_ovov","jbia"),
 ("v_vvov","beia","t1","je","x_voov","bjia"), ("w_oovv","imbe","t2","mjea","I_voov","bjia"), ("v_oovv","imbe","c","mjae","I_voov","bjia"),
 ("v_oovo","imbj","t1","ma","I_voov","bjia"), ("v_oovv","micb","t1","ma","I_vovv_p","ciab"), ("v_ovov","maic","t1","mb","I_vovv_p","ciab"),
 ("t2","jkef","v_vvov","efia","I_ooov_p","jkia"), ("t1","je","x_voov","ekia","I_ooov_p","jkia"),
 ("t1","ie","I_vv","ea","r1","ia"), ("I_oo_p","im","t1","ma","r1","ia"), ("asym","miea","I_vo","em","r1","ia"),
 ("v_oovv","miea","t1","me","r1","ia"), ("v_ovov","maie","t1","me","r1","ia"), ("v_oovo","mien","asym","mnea","r1","ia"),
 ("asym","mief","v_vvov","efma","r1","ia"),
 ("t2","ijae","I_vv","eb","r2","ijab"), ("t2","miba","I_oo","jm","r2","ijab"), ("c","ijef","v_vvvv","efab","r2","ijab"),
 ("I_oooo","ijmn","c","mnab","r2","ijab"), ("t2","mjae","I_ovov","iemb","r2","ijab"), ("I_ovov","iema","t2","mjeb","r2","ijab"),
 ("asym","miea","I_voov","ejmb","r2","ijab"), ("t1","ie","I_vovv_p","ejab","r2","ijab"), ("t1","ma","I_ooov_p","ijmb","r2","ijab"),
]
eng = Engine(0)
tot = 0.0
for (A, la, B, lb, C, lc) in sites:
    dA = tuple(D[ch] for ch in shape[A]); dB = tuple(D[ch] for ch in shape[B]); dC = tuple(D[ch] for ch in shape[C])
    ext = {}
    for l, d in list(zip(la, dA)) + list(zip(lb, dB)): ext[l] = d
    fl = 2.0
    for l in set(la + lb): fl *= ext[l]
    ms = eng.bench_contract(la, dA, lb, dB, lc, dC, reps=3)
    tot += ms
    print(f"{A+'['+la+']':16s} {B+'['+lb+']':16s} -> {C+'['+lc+']':16s} {ms:8.3f} ms {fl/ms/1e9:7.2f} TF", flush=True)
print("sum of contraction times: %.3f ms" % tot)
eng.close()
