// Minimal stand-in for <pclomp/ndt_omp.h> (TEST ONLY).
#pragma once
#include "../third_party_stub.h"
namespace pclomp {
enum NeighborSearchMethod { KDTREE, DIRECT26, DIRECT7, DIRECT1 };
template <typename PointSource, typename PointTarget>
class NormalDistributionsTransform : public hgs_test::StubEngine<PointSource, PointTarget> {
public:
  using Ptr = std::shared_ptr<NormalDistributionsTransform<PointSource, PointTarget>>;
  void setNeighborhoodSearchMethod(NeighborSearchMethod m) { search_method = m; }
  NeighborSearchMethod search_method = DIRECT7;
};
}  // namespace pclomp
