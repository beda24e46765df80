#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <omp.h>
static inline float u2f(uint32_t u){float f;memcpy(&f,&u,4);return f;}
static inline uint32_t f2u(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static inline float divc(float a,float c,float y){ float q0=a*y; float r0=fmaf(-q0,c,a); float q1=fmaf(r0,y,q0); float r1=fmaf(-q1,c,a); return fmaf(r1,y,q1);}
static inline float divc1(float a,float c,float y){ float q0=a*y; float r0=fmaf(-q0,c,a); return fmaf(r0,y,q0);}
int main(){
  // exhaustive for c = 0.001f over |a| in [2^-100, 2^100]
  float c=0.001f; float y=(float)(1.0/(double)c);
  long bad2=0,bad1=0,n=0;
  #pragma omp parallel for reduction(+:bad2,bad1,n) schedule(dynamic,1<<20)
  for(uint64_t u=0;u<(1ull<<32);++u){ float a=u2f((uint32_t)u); float fa=fabsf(a); if(!(fa>=0x1p-100f && fa<=0x1p100f)) continue; n++;
    float t=a/c; if(f2u(divc(a,c,y))!=f2u(t)) bad2++; if(f2u(divc1(a,c,y))!=f2u(t)) bad1++; }
  printf("eps: tested %ld, mismatches 2-step %ld, 1-step %ld\n",n,bad2,bad1);
  // widths
  long badw=0,badw1=0; 
  #pragma omp parallel for reduction(+:badw,badw1) schedule(dynamic,16)
  for(int W=2;W<=12000;++W){ float cw=(float)W; float yw=(float)(1.0/(double)cw); uint64_t s=W*0x9E3779B97F4A7C15ull+1;
    for(int i=0;i<400000;++i){ s^=s<<13; s^=s>>7; s^=s<<17; uint32_t u=(uint32_t)(s>>16); float a=u2f(u); float fa=fabsf(a); if(!(fa>=0x1p-100f && fa<=0x1p100f)) continue;
      float t=a/cw; if(f2u(divc(a,cw,yw))!=f2u(t)) badw++; if(f2u(divc1(a,cw,yw))!=f2u(t)) badw1++; } }
  printf("widths 2..12000 x 400k random: mismatches 2-step %ld, 1-step %ld\n",badw,badw1);
  return 0; }
