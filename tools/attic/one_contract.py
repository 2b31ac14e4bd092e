#!/usr/bin/env python3
"""One labelled contraction, default launcher choices, `reps` times (for a kernel trace).  usage: one_contract.py o v "la,lb>lc" [reps]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
o, v = int(sys.argv[1]), int(sys.argv[2])
la, rest = sys.argv[3].split(","); lb, lc = rest.split(">")
dim = lambda l: [o if ch in "ijklmn" else v for ch in l]
with Engine(0) as eng:
    print(eng.bench_contract(la, dim(la), lb, dim(lb), lc, dim(lc), int(sys.argv[4]) if len(sys.argv) > 4 else 5))
