// Hardware-semantics probe (test infrastructure, not product code).
// Pins (1) the lane<->element mapping of ds_read_b64_tr_b16 and (2) the MFMA
// 32x32x16 bf16 / 32x32x2 f32 operand+accumulator layouts on gfx950, so the
// kernels in fabric_amd/csrc can rely on measured facts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

__global__ void k_tr(short* out, const int* lane_addr_units){
  __shared__ __attribute__((aligned(16))) short lds[2048];
  for(int i=threadIdx.x;i<2048;i+=64) lds[i]=(short)i;
  __syncthreads();
  int u = lane_addr_units[threadIdx.x];   // address in units of 4 shorts (8 bytes)
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + u*4));
  for(int j=0;j<4;j++) out[threadIdx.x*4+j]=v[j];
}

__global__ void k_mfma_bf16(const float* A, const float* B, float* C){
  // A: 32x16 row-major, B: 16x32 row-major, C: 32x32 row-major
  int l = threadIdx.x;
  bf16x8 a, b;
  for(int t=0;t<8;t++){ a[t]=(__bf16)A[(l&31)*16 + (l>>5)*8+t]; b[t]=(__bf16)B[((l>>5)*8+t)*32 + (l&31)]; }
  f32x16 acc; for(int r=0;r<16;r++) acc[r]=0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a,b,acc,0,0,0);
  for(int r=0;r<16;r++){ int row=(r&3)+8*(r>>2)+4*(l>>5); int col=l&31; C[row*32+col]=acc[r]; }
}
__global__ void k_mfma_f32(const float* A, const float* B, float* C){
  // A: 32x2, B: 2x32
  int l = threadIdx.x;
  float a=A[(l&31)*2+(l>>5)], b=B[(l>>5)*32+(l&31)];
  f32x16 acc; for(int r=0;r<16;r++) acc[r]=0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a,b,acc,0,0,0);
  for(int r=0;r<16;r++){ int row=(r&3)+8*(r>>2)+4*(l>>5); int col=l&31; C[row*32+col]=acc[r]; }
}

int main(){
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,0));
  printf("device %s arch %s CUs %d clock %d kHz mem %.1f GB L2 %d\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate, p.totalGlobalMem/1e9, p.l2CacheSize);
  // ---- tr probe: several address patterns
  short* dout; int* daddr; CK(hipMalloc(&dout,64*4*2)); CK(hipMalloc(&daddr,64*4));
  for(int pat=0;pat<3;pat++){
    std::vector<int> addr(64);
    for(int l=0;l<64;l++){
      if(pat==0) addr[l]=l;                    // lane-linear 8-byte pieces
      else if(pat==1) addr[l]=(l&15)*8+(l>>4); // each lane its own 64B row, group picks piece
      else addr[l]=((l&3))+ ((l>>2)&3)*4*2 + (l>>4)*64; // rows of 8 pieces (64B rows): piece (l&3), row (l>>2)&3
    }
    CK(hipMemcpy(daddr,addr.data(),256,hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_tr,1,64,0,0,dout,daddr); CK(hipDeviceSynchronize());
    std::vector<short> o(256); CK(hipMemcpy(o.data(),dout,512,hipMemcpyDeviceToHost));
    printf("TR pattern %d (addr unit=4 shorts; value v => came from unit v/4 elem v%%4)\n",pat);
    for(int l=0;l<64;l++){ printf(" lane %2d addr_unit %3d ->",l,addr[l]); for(int j=0;j<4;j++){int v=o[l*4+j]; printf(" [u%3d.e%d]",v/4,v%4);} printf("\n"); }
  }
  // ---- mfma layout check
  {
    std::vector<float> A(32*16),B(16*32),C(32*32),R(32*32);
    for(auto&x:A) x=(float)((rand()%7)-3); for(auto&x:B) x=(float)((rand()%5)-2);
    for(int i=0;i<32;i++)for(int j=0;j<32;j++){float s=0;for(int k=0;k<16;k++)s+=A[i*16+k]*B[k*32+j];R[i*32+j]=s;}
    float *dA,*dB,*dC; CK(hipMalloc(&dA,A.size()*4));CK(hipMalloc(&dB,B.size()*4));CK(hipMalloc(&dC,C.size()*4));
    CK(hipMemcpy(dA,A.data(),A.size()*4,hipMemcpyHostToDevice));CK(hipMemcpy(dB,B.data(),B.size()*4,hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_mfma_bf16,1,64,0,0,dA,dB,dC); CK(hipDeviceSynchronize());
    CK(hipMemcpy(C.data(),dC,C.size()*4,hipMemcpyDeviceToHost));
    int bad=0; for(int i=0;i<1024;i++) if(C[i]!=R[i]) bad++;
    printf("MFMA bf16 32x32x16 layout mismatches: %d\n",bad);
    for(int i=0;i<32;i++)for(int j=0;j<32;j++){float s=0;for(int k=0;k<2;k++)s+=A[i*2+k]*B[k*32+j];R[i*32+j]=s;}
    hipLaunchKernelGGL(k_mfma_f32,1,64,0,0,dA,dB,dC); CK(hipDeviceSynchronize());
    CK(hipMemcpy(C.data(),dC,C.size()*4,hipMemcpyDeviceToHost));
    bad=0; for(int i=0;i<1024;i++) if(C[i]!=R[i]) bad++;
    printf("MFMA f32 32x32x2 layout mismatches: %d\n",bad);
  }
  return 0;
}
