// frontend_api.hip — front-end (tracker) entry points of include/plviwo.h.  (filled in below)
#include "plv_ctx.hpp"

extern "C" void plv_frontend_destroy(plv_ctx *) {}
