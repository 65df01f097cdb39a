// Layout check of v_mfma_f64_16x16x4_f64 on the device, before scripts/pending/ndt_reduction_mfma.patch is trusted:
//   A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15], D register r = D[(lane >> 4) + 4 r][lane & 15]
// with an asymmetric integer B (a symmetric one would hide a transposed read-out), and the two uses the reduction makes of it
// (one-hot column selector; 0.5 * register r as B).   hipcc --offload-arch=gfx950 -O2 scripts/probes/mfma_f64_layout.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_probe(const double* A /* [16][4] */, const double* B /* [4][16] */, double* D /* [16][16] */, double* totals /* [16] */, const double* chunks /* [16][64] */) {
  const int lane = (int)threadIdx.x;
  const d4 zero = {0.0, 0.0, 0.0, 0.0};
  const d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(lane & 15) * 4 + (lane >> 4)], B[(lane >> 4) * 16 + (lane & 15)], zero, 0, 0, 0);
  for (int r = 0; r < 4; r++) D[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = d[r];
  // the reduction: sixteen chunk sets, one-hot selector moved by row_shr:1, then four row-sum MFMAs
  d4 acc = zero;
  unsigned sel = (lane & 15) == 0 ? 0x40000000u : 0u;
  for (int c = 0; c < 16; c++) {
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(chunks[c * 64 + lane], __hiloint2double((int)sel, 0), acc, 0, 0, 0);
    sel = __builtin_amdgcn_update_dpp(0u, sel, 0x111, 0xf, 0xf, true);
  }
  d4 t = zero;
  for (int r = 0; r < 4; r++) t = __builtin_amdgcn_mfma_f64_16x16x4f64(0.5, acc[r], t, 0, 0, 0);
  if (lane < 16) totals[lane] = t[0];
}

int main() {
  double hA[64], hB[64], hD[256], hC[16 * 64], hT[16];
  srand(7);
  for (int i = 0; i < 64; i++) hA[i] = (double)(rand() % 1000 - 500), hB[i] = (double)(rand() % 1000 - 500);
  for (int i = 0; i < 16 * 64; i++) hC[i] = (double)((long long)rand() * 32768ll + rand()) * (rand() & 1 ? 1.0 : -1.0);  // |.| < 2^46
  double *dA, *dB, *dD, *dC, *dT;
  if (hipMalloc(&dA, sizeof hA) != hipSuccess) return puts("no device"), 2;
  (void)hipMalloc(&dB, sizeof hB), (void)hipMalloc(&dD, sizeof hD), (void)hipMalloc(&dC, sizeof hC), (void)hipMalloc(&dT, sizeof hT);
  (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice), (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice), (void)hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, dT, dC);
  (void)hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost), (void)hipMemcpy(hT, dT, sizeof hT, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 16; i++)
    for (int j = 0; j < 16; j++) {
      double ref = 0;
      for (int k = 0; k < 4; k++) ref += hA[i * 4 + k] * hB[k * 16 + j];
      if (ref != hD[i * 16 + j] && bad++ < 5) printf("D[%d][%d] = %.0f, expected %.0f\n", i, j, hD[i * 16 + j], ref);
    }
  for (int c = 0; c < 16; c++) {
    long long ref = 0;
    for (int l = 0; l < 64; l++) ref += (long long)hC[c * 64 + l];
    if ((double)ref != hT[c] && bad++ < 10) printf("total[%d] = %.0f, expected %lld\n", c, hT[c], ref);
  }
  puts(bad ? "MFMA f64 layout / reduction: MISMATCH" : "MFMA f64 layout / reduction: OK");
  return bad ? 1 : 0;
}
