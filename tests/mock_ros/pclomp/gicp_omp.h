// Minimal stand-in for <pclomp/gicp_omp.h> (TEST ONLY).
#pragma once
#include "../third_party_stub.h"
HGS_TEST_STUB_ENGINE(pclomp, GeneralizedIterativeClosestPoint)
