// The product links exactly one backend: the HIP kernels.  There is no runtime switch and no CPU
// fallback; without a usable GPU pga_create() fails and the error is reported through pg_last_error().
#include "pg_internal.hpp"
namespace pgx { const pga_backend_t *backend_default() { return pga_backend(); } }
