#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#if !defined(__HIP_DEVICE_COMPILE__)
__host__ __device__ static inline float hsqrt(float x){return sqrtf(x);} __host__ __device__ static inline float hdiv(float a,float b){return a/b;}
#define __fsqrt_rn hsqrt
#define __fdiv_rn hdiv
#endif
#include "qr_dev.inc"
__global__ void k(const float* A, float* out, int n){ int t=blockIdx.x*blockDim.x+threadIdx.x; if(t>=n) return; float a[5][3], b[5], x[3]; for(int j=0;j<5;j++){for(int c=0;c<3;c++) a[j][c]=A[t*15+j*3+c]; b[j]=-1.f;} qr_solve_5x3(a,b,x); float s=x[0]*x[0]+x[1]*x[1]+x[2]*x[2]; out[t*8+0]=x[0];out[t*8+1]=x[1];out[t*8+2]=x[2]; out[t*8+3]=sqrtf(s); out[t*8+4]=__fdiv_rn(x[0],out[t*8+3]); out[t*8+5]=s; out[t*8+6]=a[0][0]; out[t*8+7]=a[1][1];}
int main(){ int n=100000; std::mt19937 g(1); std::uniform_real_distribution<float> U(-1,1); std::vector<float> A(n*15), ref(n*8), got(n*8);
 for(int t=0;t<n;t++){ float nx=U(g),ny=U(g),nz=U(g); float nn=sqrtf(nx*nx+ny*ny+nz*nz); nx/=nn;ny/=nn;nz/=nn; float d0=2+14*(U(g)+1);
  float a[5][3], b[5], x[3];
  for(int j=0;j<5;j++){ float p=U(g)*2,q=U(g)*2; a[j][0]=-d0*nx + p*ny + q*nz; a[j][1]=-d0*ny - p*nx + 0.3f*q; a[j][2]=-d0*nz + 0.01f*U(g) - q*nx; for(int c=0;c<3;c++) A[t*15+j*3+c]=a[j][c]; b[j]=-1.f;}
  qr_solve_5x3(a,b,x); float s=x[0]*x[0]+x[1]*x[1]+x[2]*x[2]; ref[t*8]=x[0];ref[t*8+1]=x[1];ref[t*8+2]=x[2];ref[t*8+3]=sqrtf(s);ref[t*8+4]=x[0]/ref[t*8+3];ref[t*8+5]=s;ref[t*8+6]=a[0][0];ref[t*8+7]=a[1][1]; }
 float *dA,*dO; hipMalloc(&dA,A.size()*4); hipMalloc(&dO,got.size()*4); hipMemcpy(dA,A.data(),A.size()*4,hipMemcpyHostToDevice);
 hipLaunchKernelGGL(k,dim3((n+255)/256),dim3(256),0,0,dA,dO,n); hipMemcpy(got.data(),dO,got.size()*4,hipMemcpyDeviceToHost);
 int bad[8]={0}; int shown=0; for(int t=0;t<n;t++) for(int c=0;c<8;c++) if(memcmp(&ref[t*8+c],&got[t*8+c],4)){ bad[c]++; if(shown<6){printf("t=%d c=%d ref=%.9g got=%.9g\n",t,c,ref[t*8+c],got[t*8+c]);shown++;} }
 printf("bad: %d %d %d %d %d %d %d %d of %d\n",bad[0],bad[1],bad[2],bad[3],bad[4],bad[5],bad[6],bad[7],n); return 0; }
