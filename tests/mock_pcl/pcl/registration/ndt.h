// Minimal stand-in for <pcl/registration/ndt.h> (TEST ONLY).
#pragma once
#include <third_party_stub.h>
HGS_TEST_STUB_ENGINE(pcl, NormalDistributionsTransform)
