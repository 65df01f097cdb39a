// Minimal stand-in for <pcl/filters/passthrough.h> (TEST ONLY): see filter.h.
#pragma once
#include "filter.h"
