// Minimal stand-in for <pcl/filters/approximate_voxel_grid.h> (TEST ONLY): see filter.h.
#pragma once
#include "filter.h"
