// ABI bookkeeping entry points of libctrhip.
#include "ctr_common.h"

extern "C" int ctr_version(void) { return 33; }

extern "C" const char* ctr_target_arch(void) { return "gfx950"; }

extern "C" const char* ctr_strerror(int code) {
  switch (code) {
    case CTR_OK: return "ok";
    case CTR_EINVAL: return "invalid argument";
    case CTR_ELIMIT: return "shape outside the limits the kernels are built for";
    case CTR_ELAUNCH: return "kernel launch failed";
    case CTR_EALIGN: return "pointer or leading dimension not aligned as required";
    default: return "unknown error";
  }
}
