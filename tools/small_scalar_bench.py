import sys, time, ctypes as C, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import torch, algebra_amd as A
from algebra_amd import curves as cv
from algebra_amd._lib import check, lib
import bench
L=lib(); cid=cv.curve_id("BLS12_381_G1"); n=1<<20; ab=cv.affine_bytes(cid)
gen=np.zeros(cv.affine_words(cid),dtype=np.uint64); check(L.ark_hip_curve_generator(cid, gen.ctypes.data_as(C.c_void_p)),"g")
mul_gen=lambda k: A.into_affine(cid, A.msm_bigint(cid, gen.reshape(1,-1), bench.limbs4(k%bench.R_MOD).reshape(1,4)))
bases=torch.zeros(n*ab,dtype=torch.uint8,device="cuda"); bases[:ab]=torch.from_numpy(mul_gen(12345).view(np.uint8)).cuda(); torch.cuda.synchronize()
m=1
while m<n:
    cnt=min(m,n-m); d=np.ascontiguousarray(mul_gen(m*777)); check(L.ark_hip_sw_add_affine_device(cid,bases.data_ptr(),bases.data_ptr()+m*ab,cnt,d.ctypes.data_as(C.c_void_p)),"e"); m+=cnt
rng=np.random.default_rng(1)
def run(name, vals):
    sc=np.zeros((n,4),dtype=np.uint64); sc[:,0]=vals
    s=torch.from_numpy(sc.view(np.int64)).cuda(); torch.cuda.synchronize()
    A.msm_bigint(cid,bases,s); t=time.perf_counter()
    for _ in range(3): A.msm_bigint(cid,bases,s)
    print("%-10s 2^20: %.2f ms" % (name,(time.perf_counter()-t)/3*1e3))
run("bool", rng.integers(0,2,size=n,dtype=np.uint64))
run("u8", rng.integers(0,256,size=n,dtype=np.uint64))
run("u16", rng.integers(0,1<<16,size=n,dtype=np.uint64))
run("u32", rng.integers(0,1<<32,size=n,dtype=np.uint64))
run("u64", rng.integers(0,1<<63,size=n,dtype=np.uint64))
run("all-equal", np.full(n, 0x1234567, dtype=np.uint64))
