// Same name as the reference header; the declarations live in jpegenc_host.h.
#pragma once
#include "jpegenc_host.h"
