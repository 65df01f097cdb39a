// Minimal stand-in for <boost/bind.hpp> (TEST ONLY): boost::bind with the global placeholders _1, _2 (roscpp's headers bring both in).
#pragma once
#include <functional>
namespace boost {
using std::bind;
}  // namespace boost
using std::placeholders::_1;
using std::placeholders::_2;
