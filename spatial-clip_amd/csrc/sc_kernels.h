// Internal alias of the public C ABI header (include/spatial_clip_hip.h).
#pragma once
#include "../../include/spatial_clip_hip.h"
