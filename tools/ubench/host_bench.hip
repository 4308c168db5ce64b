#include "../../c-kzg-4844_amd/csrc/host_pairing.hpp"
#include <chrono>
#include <cstdio>
using namespace ckzg; using namespace ckzg::host;
int main(){
  G1Jac g=g1_generator(); Fp a=g.x,b=g.y;
  auto t0=std::chrono::steady_clock::now();
  for(int i=0;i<2000000;i++){ a=mul(a,b);} auto t1=std::chrono::steady_clock::now();
  printf("fp mul %.1f ns (%x)\n", std::chrono::duration<double,std::nano>(t1-t0).count()/2e6, a.l[0]);
  t0=std::chrono::steady_clock::now();
  for(int i=0;i<2000000;i++){ a=add(a,b); b=sub(b,a);} t1=std::chrono::steady_clock::now();
  printf("fp add+sub %.1f ns (%x)\n", std::chrono::duration<double,std::nano>(t1-t0).count()/2e6, a.l[0]);
  Fp12 f=Fp12::one(); f.c0.c1.c0=a; f.c1.c2.c1=b; f.c1.c0.c0=g.x;
  t0=std::chrono::steady_clock::now();
  for(int i=0;i<20000;i++){ f=mul(f,f);} t1=std::chrono::steady_clock::now();
  printf("fp12 mul %.2f us (%x)\n", std::chrono::duration<double,std::micro>(t1-t0).count()/2e4, f.c0.c0.c0.l[0]);
  G2Prepared p1,p2; g2_prepare(p1,g2_to_affine(g2_generator())); g2_prepare(p2,g2_to_affine(g2_dbl(g2_generator())));
  G1Affine x=jac_to_affine(g), y=jac_to_affine(jac_dbl(g));
  t0=std::chrono::steady_clock::now(); bool r=false;
  for(int i=0;i<10;i++){ r^=pairing_product_is_one(x,p2,affine_neg(y),p1);} t1=std::chrono::steady_clock::now();
  printf("pairing product %.2f ms (%d)\n", std::chrono::duration<double,std::milli>(t1-t0).count()/10, r);
}
