#pragma once
#include "embree3/rtcore.h"
