// Check of the sinf/cosf restatement used on the device against the host libm.
// TEST INFRASTRUCTURE ONLY.  The HIP descriptor kernel evaluates the steering
// sine/cosine with this algorithm (glibc 2.28+ sinf/cosf: double-precision
// reduction by pi/2, minimax polynomials, one rounding to float).  This program
// compares it with the libm the reference would call (reference
// src/ORBextractor.cc:97) for every `stride`-th float in [0, 6.5]; stride 1 is
// exhaustive (1.09e9 values, ~12 s).  Build with -DUSE_FMA -mfma to check the
// fused form.  Usage: check_sincosf [stride]
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
typedef struct { double sign[4]; double hpi_inv, hpi, c0,c1,c2,c3,c4, s1,s2,s3; } sincos_t;
static const sincos_t T[2] = {
 {{1.0,-1.0,-1.0,1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
 {{1.0,-1.0,-1.0,1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};
#ifdef USE_FMA
#define MADD(a,b,c) fma((a),(b),(c))
#else
#define MADD(a,b,c) ((a)*(b)+(c))
#endif
static inline float sinf_poly(double x, double x2, const sincos_t*p, int n){
  double x3,x4,x6,x7,s,c,c1,c2,s1;
  if((n&1)==0){ x3=x*x2; s1=MADD(x2,p->s3,p->s2); x7=x3*x2; s=MADD(x3,p->s1,x); return (float)MADD(x7,s1,s); }
  else { x4=x2*x2; c2=MADD(x2,p->c4,p->c3); c1=MADD(x2,p->c2,p->c1); x6=x4*x2; c=MADD(x2,c1,p->c0); return (float)MADD(x6,c2,c); }
}
static inline double reduce_fast(double x,const sincos_t*p,int*np){
  double r=x*p->hpi_inv; int n=((int32_t)r+0x800000)>>24; *np=n; return MADD(-(double)n, p->hpi, x);
}
static inline uint32_t asu(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static inline uint32_t abstop12(float x){return (asu(x)>>20)&0x7ff;}
static float my_sinf(float y){
  double x=y,s; int n; const sincos_t*p=&T[0];
  if(abstop12(y)<abstop12(0x1.921FB6p-1f)){ s=x*x; if(abstop12(y)<abstop12(0x1p-12f)) return y; return sinf_poly(x,s,p,0);}
  x=reduce_fast(x,p,&n); s=p->sign[n&3]; if(n&2)p=&T[1]; return sinf_poly(x*s,x*x,p,n);
}
static float my_cosf(float y){
  double x=y,s; int n; const sincos_t*p=&T[0];
  if(abstop12(y)<abstop12(0x1.921FB6p-1f)){ s=x*x; if(abstop12(y)<abstop12(0x1p-12f)) return 1.0f; return sinf_poly(x,s,p,1);}
  x=reduce_fast(x,p,&n); s=p->sign[(n+1)&3]; if((n+1)&2)p=&T[1]; return sinf_poly(x*s,x*x,p,n^1);
}
int main(int argc,char**argv){
  long bad_s=0,bad_c=0,tot=0; uint32_t first=0; uint32_t stride = argc>1 ? (uint32_t)atoi(argv[1]) : 1; if(!stride) stride=1;
  for(uint32_t u=0; u<=asu(6.5f); u+=stride){ float f; memcpy(&f,&u,4); float a=sinf(f),b=my_sinf(f); float c=cosf(f),d=my_cosf(f);
    if(asu(a)!=asu(b)){ if(!bad_s)first=u; bad_s++;} if(asu(c)!=asu(d))bad_c++; tot++; }
  printf("total %ld bad_sin %ld bad_cos %ld first %08x\n",tot,bad_s,bad_c,first);
  return (bad_s||bad_c) ? 1 : 0; }
