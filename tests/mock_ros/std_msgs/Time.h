// Minimal stand-in for <std_msgs/Time.h> (TEST ONLY).
#pragma once
#include "Header.h"
