// FETCH_SIZE calibration on gfx950 for the two read widths of conv_split_kernel (MI355X_MICROARCH.md §HBM: "FETCH_SIZE reports
// exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) ... other access widths are uncalibrated: calibrate on a
// known byte count in your own access pattern").  Each kernel streams N bytes of a buffer larger than the 256 MB Infinity Cache once:
//   read16: one 16-B load per lane (the operand staging's width)      read4: one dword per lane (the staged epilogue's width)
//   read4_planes: dword per lane, 8 channel planes per thread as the epilogue's AS_EPI_LOADK issues them (plane stride = pixels * 4)
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/experiments/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/fc -- /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void read16(const float4* __restrict__ p, float* __restrict__ out, long long n16) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.f;
  for (; i < n16; i += (long long)gridDim.x * blockDim.x) { const float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 12345.678f) out[0] = acc;
}
__global__ void read4(const float* __restrict__ p, float* __restrict__ out, long long n4) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.f;
  for (; i < n4; i += (long long)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 12345.678f) out[0] = acc;
}
__global__ void read4_planes(const float* __restrict__ p, float* __restrict__ out, long long pixels, int planes) {
  const long long px = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (px >= pixels) return;
  float acc = 0.f;
  for (int c0 = 0; c0 < planes; c0 += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p[(long long)(c0 + j) * pixels + px];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j];
  }
  if (acc == 12345.678f) out[0] = acc;
}

int main() {
  const long long bytes = 1ll << 30;  // 1 GiB: four times the Infinity Cache
  float *buf, *out;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
  hipMemset(buf, 0, bytes);
  hipDeviceSynchronize();
  for (int r = 0; r < 3; ++r) {
    hipLaunchKernelGGL(read16, dim3(4096), dim3(256), 0, 0, (const float4*)buf, out, bytes / 16);
    hipLaunchKernelGGL(read4, dim3(4096), dim3(256), 0, 0, buf, out, bytes / 4);
    const long long pixels = 1ll << 20;  // 256 planes of 1 Mi pixels
    hipLaunchKernelGGL(read4_planes, dim3((unsigned)(pixels / 256)), dim3(256), 0, 0, buf, out, pixels, 256);
  }
  hipDeviceSynchronize();
  printf("each kernel read %lld bytes (%.1f MB)\n", bytes, bytes / 1e6);
  return 0;
}
