// Probe: buffer_load_dwordx4 ... lds on gfx950 -- destination order and what out-of-range lanes write.
// build: hipcc --offload-arch=gfx950 -O2 tools/probe/glds_probe.hip -o tools/probe/glds_probe ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const void* p, unsigned bytes, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (int i = threadIdx.x; i < 2048 / 4; i += blockDim.x) ((unsigned*)smem)[i] = 0xdeadbeefu;
  __syncthreads();
  auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
  unsigned off = (63 - threadIdx.x) * 16;                 // reversed source order
  if ((threadIdx.x & 3) == 1) off = 0x80000000u;          // out of range
  if ((threadIdx.x & 3) == 2) off = bytes;                // just past the end
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 512), 16, off, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 2048 / 4; i += blockDim.x) out[i] = ((unsigned*)smem)[i];
}
int main() {
  unsigned h[256], *d, *o, ho[512];
  for (int i = 0; i < 256; ++i) h[i] = 0x1000 + i;        // 64 chunks of 16 B
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, (unsigned)sizeof(h), o);
  hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
  for (int lane = 0; lane < 8; ++lane) printf("lane %d -> lds word %d: %08x %08x %08x %08x\n", lane, 128 + lane * 4, ho[128 + lane * 4], ho[129 + lane * 4], ho[130 + lane * 4], ho[131 + lane * 4]);
  printf("before dest: %08x  after dest: %08x\n", ho[127], ho[128 + 256]);
  return 0;
}
