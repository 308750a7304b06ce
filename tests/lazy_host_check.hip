// Host-side check of the 28-bit-limb ("lazy") field and mixed addition (algebra_amd/csrc/ubench/lazy.cuh) against the
// saturated form (fp.cuh / ec.cuh, itself checked against the oracle): conversions, products, lazy add/sub,
// exact zero tests, and random point sequences that hit the doubling and infinity branches.  The templates are
// __host__ __device__, so this runs on the CPU.  Built and run by tests/test_lazy_host.py.
#include "lazy.cuh"
#include "curves.cuh"
#include <stdio.h>
#include <random>
using namespace arkhip;
template<class P> Fp<P> rnd(std::mt19937_64& g){ Fp<P> a; for(int i=0;i<P::N;i++) a.l[i]=(u32)g(); a.l[P::N-1]&= (1u<<((P::BITS-1)%32))-1; return a; }
template<class P> int run(const char* name, const uint64_t* gen){
  typedef Fp<P> F; typedef FpLazy<P> L;
  std::mt19937_64 g(7); int bad=0;
  for(int it=0;it<2000;it++){
    F a=rnd<P>(g), b=rnd<P>(g);
    L la=L::from_canonical(a), lb=L::from_canonical(b);
    if(!F::eq(la.to_canonical(), a)) {bad++; if(bad<5) printf("%s roundtrip fail\n",name);}
    if(!F::eq(L::mul(la,lb).to_canonical(), F::mul(a,b))) {bad++; if(bad<5) printf("%s mul fail\n",name);}
    if(!F::eq(L::template sub<1>(la,lb).to_canonical(), F::sub(a,b))) {bad++; if(bad<5) printf("%s sub fail\n",name);}
    if(!F::eq(L::add_lazy(la,lb).to_canonical(), F::add(a,b))) {bad++; if(bad<5) printf("%s add fail\n",name);}
    L z=L::template sub<3>(la,la); if(!z.is_zero_mod_p()) {bad++; printf("zero test fail\n");}
    if(L::template sub<2>(la,lb).is_zero_mod_p() && !F::eq(a,b)) {bad++; printf("false zero\n");}
  }
  // point sequences: start from generator G, build multiples by madd both ways
  F gx=F::load(gen), gy=F::load((const char*)gen+F::BYTES);
  // points: P_k = k*G affine via canonical arithmetic (xyzz then normalise using inverse)
  const int NP=40; F px[NP], py[NP];
  XYZZ<F> acc=XYZZ<F>::zero();
  for(int k=0;k<NP;k++){ xyzz_madd<F>(acc,gx,gy); F zi=F::inverse(acc.zzz); F zzi=F::sqr(F::mul(acc.zz,zi)); px[k]=F::mul(acc.x,zzi); py[k]=F::mul(acc.y,zi); }
  for(int trial=0;trial<200;trial++){
    XYZZ<F> c=XYZZ<F>::zero(); XYZZLazy<P> lz; lz.inf=true; lz.x=lz.y=lz.zz=lz.zzz=L::zero();
    int len = 1 + g()%12;
    for(int s=0;s<len;s++){
      int k=g()%NP; bool neg=g()&1;
      if(trial%5==0 && s==1) { /* force same point twice or inverse */ }
      F y=F::cond_neg(py[k],neg);
      xyzz_madd<F>(c,px[k],y);
      xyzz_madd_lazy<P>(lz, L::from_canonical(px[k]), L::from_canonical(py[k]), neg);
      if(trial%3==0){ // repeat the same point: doubling; then its inverse twice
        xyzz_madd<F>(c,px[k],y); xyzz_madd_lazy<P>(lz, L::from_canonical(px[k]), L::from_canonical(py[k]), neg);
      }
      if(trial%7==0){ F ny=F::neg(y); xyzz_madd<F>(c,px[k],ny); xyzz_madd_lazy<P>(lz, L::from_canonical(px[k]), L::from_canonical(py[k]), !neg); }
    }
    XYZZ<F> d=xyzz_from_lazy<P>(lz);
    // compare as group elements: cross-multiply
    bool same;
    if(c.is_zero()||d.is_zero()) same = c.is_zero()&&d.is_zero();
    else same = F::eq(F::mul(c.x,d.zz),F::mul(d.x,c.zz)) && F::eq(F::mul(c.y,d.zzz),F::mul(d.y,c.zzz));
    if(!same){bad++; if(bad<8) printf("%s point seq mismatch trial %d\n",name,trial);}
  }
  printf("%s: %s (%d bad)\n", name, bad?"FAIL":"ok", bad);
  return bad;
}
#include "curve_consts.hpp"
int main(){ int b=0; b+=run<BLS12_381_FQ>("BLS12_381_FQ",GEN_BLS12_381_G1); b+=run<BN254_FQ>("BN254_FQ",GEN_BN254_G1); b+=run<BLS12_377_FQ>("BLS12_377_FQ",GEN_BLS12_377_G1); return b!=0; }
