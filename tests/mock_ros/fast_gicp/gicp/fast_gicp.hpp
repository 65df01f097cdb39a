// Minimal stand-in for <fast_gicp/gicp/fast_gicp.hpp> (TEST ONLY).
#pragma once
#include "../../third_party_stub.h"
HGS_TEST_STUB_ENGINE(fast_gicp, FastGICP)
