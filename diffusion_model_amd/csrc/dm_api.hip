// ABI identity.
#include "dm_common.h"
extern "C" int dm_abi_version(void) { return 1; }
extern "C" const char* dm_target_arch(void) { return "gfx950"; }
