// Minimal stand-in for <fast_gicp/gicp/fast_vgicp.hpp> (TEST ONLY).
#pragma once
#include "../../third_party_stub.h"
HGS_TEST_STUB_ENGINE(fast_gicp, FastVGICP)
