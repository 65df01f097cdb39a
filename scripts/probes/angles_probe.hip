// Probe: are ndt_angle_tables / ndt_point_derivatives bit-identical on host and device? (scripts/probes, not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "../../hdl_graph_slam_amd/csrc/hgs_ndt.h"
using namespace hgs;
__global__ void k(const double* p, NdtAngles* out, double* sc, NdtPointDeriv* pd) {
  ndt_angle_tables(p, 1, *out);
  for (int i = 0; i < 3; i++) sc[2 * i] = sin(p[3 + i]), sc[2 * i + 1] = cos(p[3 + i]);
  ndt_point_derivatives(*out, 12.345f, -3.21f, 0.777f, *pd);
}
int main() {
  double p[6] = {1.5, 0.2, 0.0, 3.14, 3.13, 3.1};
  double* dp; NdtAngles* da; double* dsc; NdtPointDeriv* dpd;
  hipMalloc(&dp, 48); hipMalloc(&da, sizeof(NdtAngles)); hipMalloc(&dsc, 48); hipMalloc(&dpd, sizeof(NdtPointDeriv));
  hipMemcpy(dp, p, 48, hipMemcpyHostToDevice);
  k<<<1, 1>>>(dp, da, dsc, dpd);
  NdtAngles ga, ha; double gsc[6]; NdtPointDeriv gpd, hpd;
  hipMemcpy(&ga, da, sizeof(ga), hipMemcpyDeviceToHost); hipMemcpy(gsc, dsc, 48, hipMemcpyDeviceToHost); hipMemcpy(&gpd, dpd, sizeof(gpd), hipMemcpyDeviceToHost);
  ndt_angle_tables(p, 1, ha);
  ndt_point_derivatives(ha, 12.345f, -3.21f, 0.777f, hpd);
  for (int i = 0; i < 3; i++) printf("sin %a %a  cos %a %a\n", gsc[2 * i], sin(p[3 + i]), gsc[2 * i + 1], cos(p[3 + i]));
  int nd = 0;
  const float* a = (const float*)&ga; const float* b = (const float*)&ha;
  for (size_t i = 0; i < sizeof(ga) / 4; i++) if (memcmp(a + i, b + i, 4)) { printf("angles[%zu] dev %a host %a\n", i, a[i], b[i]); nd++; }
  a = (const float*)&gpd; b = (const float*)&hpd;
  for (size_t i = 0; i < sizeof(gpd) / 4; i++) if (memcmp(a + i, b + i, 4)) { printf("pd[%zu] dev %a host %a\n", i, a[i], b[i]); nd++; }
  printf("differences: %d\n", nd);
  return 0;
}
