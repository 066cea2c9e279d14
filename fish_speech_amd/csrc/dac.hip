// placeholder until the codec lands
#include "dac_kernels.h"
using namespace fmi;
struct fmi_dac { int dummy; };
extern "C" {
int64_t fmi_dac_arena_bytes(const fmi_dac_config*) { return -1; }
int fmi_dac_create(const fmi_dac_config*, void*, int64_t, fmi_dac**) { return set_error(FMI_ESTATE, "codec not built"); }
void fmi_dac_destroy(fmi_dac*) {}
int fmi_dac_load_tensor(fmi_dac*, const char*, const float*, int, const int64_t*, int, void*) { return set_error(FMI_ESTATE, "codec not built"); }
int fmi_dac_finalize_weights(fmi_dac*, void*) { return set_error(FMI_ESTATE, "codec not built"); }
int fmi_dac_weights_ready(fmi_dac*) { return set_error(FMI_ESTATE, "codec not built"); }
int fmi_dac_decode(fmi_dac*, int64_t*, int, int, float*, void*) { return set_error(FMI_ESTATE, "codec not built"); }
int fmi_dac_encode(fmi_dac*, const float*, int, int, int64_t*, void*) { return set_error(FMI_ESTATE, "codec not built"); }
int fmi_dac_frame_length(const fmi_dac*) { return 0; }
int fmi_dac_debug_z(fmi_dac*, float**, int*, int*) { return set_error(FMI_ESTATE, "codec not built"); }
}
