// holoscene_amd/csrc/capi.hip -- library-level entry points of libholoscene_hip.so.
#include "holoscene_hip.h"

extern "C" {

int hs_abi_version(void) { return 9; }

const char *hs_target_arch(void) { return "gfx950"; }

}  // extern "C"
