#pragma once
#include "time.h"
