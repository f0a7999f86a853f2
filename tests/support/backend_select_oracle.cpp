// TEST INFRASTRUCTURE ONLY: links the host driver against the plain-C oracle backend so that the
// reader, the writers and the round driver can be checked against the reference on a machine without
// a GPU.  Never part of libpangene_amd.so.
#include "pangene_hip.h"
extern "C" const pga_backend_t *pgo_backend(void);
namespace pgx { const pga_backend_t *backend_default() { return pgo_backend(); } }
