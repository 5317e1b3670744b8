// Hardware probe: verifies the MFMA fragment-layout and ds_read_b64_tr_b16 assumptions the kernels
// in transfusion_pytorch_amd/csrc rely on.  Build: hipcc --offload-arch=gfx950 -O2 probe_layouts.hip -o probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

__device__ inline __bf16 f2bf(float f){ return (__bf16)f; }

// (a) C[32][32] = A[32][16] * Bt[32][16]^T with hypothesised layouts
__global__ void k_mfma32(const float* A, const float* Bt, float* C){
  int l = threadIdx.x;
  bf16x8 a, b;
  for(int e=0;e<8;e++){ a[e]=f2bf(A[(l&31)*16 + 8*(l>>5)+e]); b[e]=f2bf(Bt[(l&31)*16 + 8*(l>>5)+e]); }
  f32x16 acc; for(int i=0;i<16;i++) acc[i]=0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a,b,acc,0,0,0);
  for(int r=0;r<16;r++){ int row=(r&3)+8*(r>>2)+4*(l>>5); int col=l&31; C[row*32+col]=acc[r]; }
}
// (b) C[16][16] = A[16][32] * Bt[16][32]^T
__global__ void k_mfma16(const float* A, const float* Bt, float* C){
  int l = threadIdx.x;
  bf16x8 a, b;
  for(int e=0;e<8;e++){ a[e]=f2bf(A[(l&15)*32 + 8*(l>>4)+e]); b[e]=f2bf(Bt[(l&15)*32 + 8*(l>>4)+e]); }
  f32x4 acc={0,0,0,0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a,b,acc,0,0,0);
  for(int r=0;r<4;r++){ int row=(l>>4)*4+r; int col=l&15; C[row*16+col]=acc[r]; }
}
// (c) raw tr-read dump: LDS holds shorts 0..4095; lane address = base(l) supplied by host table (in shorts)
__global__ void k_tr(const int* addr, short* out){
  __shared__ __attribute__((aligned(16))) short lds[4096];
  int l=threadIdx.x;
  for(int i=l;i<4096;i+=64) lds[i]=(short)i;
  __syncthreads();
  s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + addr[l]));
  for(int e=0;e<4;e++) out[l*4+e]=t[e];
}
// (d) intended usage: X[kk][n] row-major (stride RS shorts) in LDS; want lane l to get the 8 values
// X[8*(l>>5)+0..7][l&31] (A operand of 32x32x16 from a [16][32] k-major tile) using two tr reads.
__global__ void k_tr_use(short* out, int RS){
  __shared__ __attribute__((aligned(16))) short lds[4096];
  int l=threadIdx.x;
  for(int i=l;i<4096;i+=64) lds[i]=(short)i;   // X[kk][n] = kk*RS+n
  __syncthreads();
  int g=l>>4, q=l&15;
  int nbase=16*(g&1), kbase=8*(g>>1);
  int a0=(kbase + (q>>2))*RS + nbase + 4*(q&3);
  int a1=(kbase + 4 + (q>>2))*RS + nbase + 4*(q&3);
  s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + a0));
  s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + a1));
  for(int e=0;e<4;e++){ out[l*8+e]=t0[e]; out[l*8+4+e]=t1[e]; }
}
int main(){
  int dev=0; CK(hipSetDevice(dev)); hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,dev));
  printf("device %s arch %s CUs %d\n", p.name, p.gcnArchName, p.multiProcessorCount);
  // (a)
  { std::vector<float> A(32*16),B(32*16),C(32*32),R(32*32);
    for(int i=0;i<32*16;i++){A[i]=(float)((i*7)%13-6); B[i]=(float)((i*5+3)%11-5);}
    for(int i=0;i<32;i++)for(int j=0;j<32;j++){float s=0;for(int k=0;k<16;k++)s+=A[i*16+k]*B[j*16+k];R[i*32+j]=s;}
    float *dA,*dB,*dC; CK(hipMalloc(&dA,2048));CK(hipMalloc(&dB,2048));CK(hipMalloc(&dC,4096));
    CK(hipMemcpy(dA,A.data(),2048,hipMemcpyHostToDevice));CK(hipMemcpy(dB,B.data(),2048,hipMemcpyHostToDevice));
    k_mfma32<<<1,64>>>(dA,dB,dC); CK(hipDeviceSynchronize()); CK(hipMemcpy(C.data(),dC,4096,hipMemcpyDeviceToHost));
    int bad=0; for(int i=0;i<1024;i++) if(fabs(C[i]-R[i])>1e-3) bad++;
    printf("PROBE mfma32x32x16 layout: %s (%d mismatches)\n", bad? "FAIL":"OK", bad);
    if(bad){ // transposed?
      int badT=0; for(int i=0;i<32;i++)for(int j=0;j<32;j++) if(fabs(C[j*32+i]-R[i*32+j])>1e-3) badT++;
      printf("  transposed mismatches %d\n", badT); }
  }
  { std::vector<float> A(16*32),B(16*32),C(256),R(256);
    for(int i=0;i<512;i++){A[i]=(float)((i*7)%13-6); B[i]=(float)((i*5+3)%11-5);}
    for(int i=0;i<16;i++)for(int j=0;j<16;j++){float s=0;for(int k=0;k<32;k++)s+=A[i*32+k]*B[j*32+k];R[i*16+j]=s;}
    float *dA,*dB,*dC; CK(hipMalloc(&dA,2048));CK(hipMalloc(&dB,2048));CK(hipMalloc(&dC,1024));
    CK(hipMemcpy(dA,A.data(),2048,hipMemcpyHostToDevice));CK(hipMemcpy(dB,B.data(),2048,hipMemcpyHostToDevice));
    k_mfma16<<<1,64>>>(dA,dB,dC); CK(hipDeviceSynchronize()); CK(hipMemcpy(C.data(),dC,1024,hipMemcpyDeviceToHost));
    int bad=0; for(int i=0;i<256;i++) if(fabs(C[i]-R[i])>1e-3) bad++;
    printf("PROBE mfma16x16x32 layout: %s (%d mismatches)\n", bad? "FAIL":"OK", bad);
  }
  // (c) raw dump with contiguous addressing: lane l -> shorts offset l*4
  { std::vector<int> addr(64); for(int l=0;l<64;l++) addr[l]=l*4;
    int* dA; short* dO; CK(hipMalloc(&dA,256)); CK(hipMalloc(&dO,512));
    CK(hipMemcpy(dA,addr.data(),256,hipMemcpyHostToDevice));
    k_tr<<<1,64>>>(dA,dO); CK(hipDeviceSynchronize());
    std::vector<short> o(256); CK(hipMemcpy(o.data(),dO,512,hipMemcpyDeviceToHost));
    printf("PROBE tr16_b64 raw (lane addr = 4*lane shorts): lane: e0 e1 e2 e3\n");
    for(int l=0;l<64;l++){ printf("  L%02d: %4d %4d %4d %4d%s", l,o[l*4],o[l*4+1],o[l*4+2],o[l*4+3], (l%4==3)?"\n":" |"); }
    int bad=0; for(int l=0;l<64;l++)for(int j=0;j<4;j++) if(o[l*4+j]!=(l&15)+j*16+(l>>4)*64) bad++;
    printf("PROBE tr16_b64 contiguous hypothesis: %s (%d mismatches)\n", bad?"FAIL":"OK", bad);
    // strided rows: each 16-lane group: row stride 40 shorts; lane q -> row q/4, chunk q%4
    for(int l=0;l<64;l++){ int g=l>>4,q=l&15; addr[l]=g*256 + (q>>2)*40 + (q&3)*4; }
    CK(hipMemcpy(dA,addr.data(),256,hipMemcpyHostToDevice));
    k_tr<<<1,64>>>(dA,dO); CK(hipDeviceSynchronize()); CK(hipMemcpy(o.data(),dO,512,hipMemcpyDeviceToHost));
    bad=0; for(int l=0;l<64;l++)for(int j=0;j<4;j++){ int g=l>>4,q=l&15; int exp=g*256 + j*40 + q; if(o[l*4+j]!=exp) bad++; }
    printf("PROBE tr16_b64 strided-row hypothesis (lane q' loads row q'/4 chunk q'%%4; lane q gets col q rows 0..3): %s (%d mismatches)\n", bad?"FAIL":"OK", bad);
    if(bad){ for(int l=0;l<64;l++){ printf("  L%02d: %4d %4d %4d %4d%s", l,o[l*4],o[l*4+1],o[l*4+2],o[l*4+3], (l%4==3)?"\n":" |"); } }
  }
  // (d)
  for(int RS : {32, 72, 136}) { short* dO; CK(hipMalloc(&dO,1024));
    k_tr_use<<<1,64>>>(dO,RS); CK(hipDeviceSynchronize());
    std::vector<short> o(512); CK(hipMemcpy(o.data(),dO,1024,hipMemcpyDeviceToHost));
    int bad=0; for(int l=0;l<64;l++)for(int e=0;e<8;e++){ int exp=(8*(l>>5)+e)*RS + (l&31); if(o[l*8+e]!=exp) bad++; }
    printf("PROBE tr-use A-operand-from-k-major (RS=%d): %s (%d mismatches)\n", RS, bad?"FAIL":"OK", bad);
  }
  return 0;
}
