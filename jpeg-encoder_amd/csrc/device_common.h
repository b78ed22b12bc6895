// Small types shared by the device translation units of libjpegenc_hip.
#ifndef JPEGHIP_DEVICE_COMMON_H
#define JPEGHIP_DEVICE_COMMON_H

#include <hip/hip_runtime.h>

namespace jpeghip {

typedef unsigned long long u64;
typedef unsigned int u32;

}  // namespace jpeghip

#endif
