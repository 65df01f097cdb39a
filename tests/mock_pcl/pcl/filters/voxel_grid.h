// Minimal stand-in for <pcl/filters/voxel_grid.h> (TEST ONLY): see filter.h.
#pragma once
#include "filter.h"
