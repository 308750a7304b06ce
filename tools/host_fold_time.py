#!/usr/bin/env python3
"""Host tail of an MSM timed alone (ark_hip_test_msm_host_fold on the part sums of a 2^16-pair layout): no GPU involved.
    [ARK_HIP_LIB=...] python tools/host_fold_time.py"""
import sys, time, ctypes as C
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import oracle_lib as O
from algebra_amd import _lib
A4 = np.array([0xA11CE, 1, 2, 0], dtype=np.uint64); B4 = np.array([0xB0B, 3, 0, 0], dtype=np.uint64)
for cname,windows,nbits,l0,widths in (("BLS12_381_G1",22,9,2,[12]*13+[11]*9),("BN254_G1",22,9,2,[12]*12+[11]*10),("BLS12_377_G2",19,11,2,[14]*6+[13]*13)):
    cid=O.CID[cname]; fw=O.fe_words(cid)
    npts=windows*(nbits+1)
    aff=O.gen_bases(cid,A4,B4,npts)
    one=O.field_const(O.curve_info(cid)[0],1)
    parts=np.zeros((windows,nbits+1,4*fw),dtype=np.uint64)
    for k in range(npts):
        w,q=divmod(k,nbits+1)
        parts[w,q,:2*fw]=aff[k]
        parts[w,q,2*fw:2*fw+one.size]=one
        parts[w,q,3*fw:3*fw+one.size]=one
    out=np.zeros(3*fw,dtype=np.uint64)
    wid=(C.c_int*windows)(*widths)
    L=_lib.test_lib()
    f=lambda: L.ark_hip_test_msm_host_fold(cid,parts.ctypes.data_as(C.c_void_p),windows,nbits,l0,wid,out.ctypes.data_as(C.c_void_p))
    f(); best=1e9
    for rep in range(5):
        t0=time.perf_counter()
        for _ in range(10): f()
        best=min(best,(time.perf_counter()-t0)/10)
    print(cname,"host fold %.1f us"%(best*1e6), hex(int(out[0])))
