// Minimal stand-in for <pcl/filters/statistical_outlier_removal.h> (TEST ONLY): see filter.h.
#pragma once
#include "filter.h"
