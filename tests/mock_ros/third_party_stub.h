// TEST ONLY: what the stand-ins for ndt_omp / fast_gicp / PCL's own engines have in common.  The factory test only needs the reference's
// untouched branches to COMPILE next to the new ones; a stub engine that is actually asked to align reports "not converged".
#pragma once
#include <pcl/point_types.h>
#include <pcl/registration/registration.h>
namespace hgs_test {
template <typename PointSource, typename PointTarget>
class StubEngine : public pcl::Registration<PointSource, PointTarget, float> {
public:
  using Base = pcl::Registration<PointSource, PointTarget, float>;
  void setNumThreads(int) {}
  void setResolution(double r) { resolution = r; }
  void setCorrespondenceRandomness(int k) { k_correspondences = k; }
  void setUseReciprocalCorrespondences(bool) {}
  void setMaximumOptimizerIterations(int) {}
  double resolution = 0;
  int k_correspondences = 0;

protected:
  void computeTransformation(typename Base::PointCloudSource&, const typename Base::Matrix4& guess) override {
    this->final_transformation_ = guess;
    this->converged_ = false;
  }
};
}  // namespace hgs_test
#define HGS_TEST_STUB_ENGINE(NS, NAME)                                   \
  namespace NS {                                                         \
  template <typename PointSource, typename PointTarget>                  \
  class NAME : public hgs_test::StubEngine<PointSource, PointTarget> {   \
  public:                                                                \
    using Ptr = std::shared_ptr<NAME<PointSource, PointTarget>>;         \
  };                                                                     \
  }
