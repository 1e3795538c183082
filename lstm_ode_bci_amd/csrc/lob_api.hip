// Library-level entry points of liblob.so.
#include "lob_common.h"

extern "C" int lob_version(void) { return LOB_VERSION; }
