// Temporary: entries declared in include/uwu_hip.h that are not implemented yet return UWU_ENOTIMPL.
#include <stddef.h>
#include <stdint.h>
#include "../../include/uwu_hip.h"
void uwu_set_error(const char* fmt, ...);
#define STUB(name) { uwu_set_error(#name ": not implemented"); return UWU_ENOTIMPL; }
extern "C" {
int uwu_colsum(const void*, int, int, int, int, float*, int, void*) STUB(uwu_colsum)
int uwu_add_ln_modulate_fwd(const void*, const void*, const float*, const float*, const float*, int, void*, void*, float*, float*, int, int, int, float, int, void*) STUB(uwu_add_ln_modulate_fwd)
int uwu_add_ln_modulate_bwd(const void*, const void*, const float*, const float*, const float*, const void*, const void*, const float*, int, void*, void*, float*, float*, float*, int, int, int, int, void*) STUB(uwu_add_ln_modulate_bwd)
int uwu_attention_fwd(const void*, void*, float*, int, int, int, int, int, void*) STUB(uwu_attention_fwd)
int uwu_attention_bwd(const void*, const void*, const void*, const float*, float*, void*, int, int, int, int, int, void*) STUB(uwu_attention_bwd)
int uwu_timestep_embedding(const float*, int, int, float, void*, int, void*) STUB(uwu_timestep_embedding)
int uwu_silu_fwd(const void*, void*, int64_t, int, void*) STUB(uwu_silu_fwd)
int uwu_silu_bwd(const void*, const void*, void*, int64_t, int, void*) STUB(uwu_silu_bwd)
int uwu_patchify(const float*, void*, int, int, int, int, int, int, void*) STUB(uwu_patchify)
int uwu_unpatchify(const void*, int, float*, int, int, int, int, int, void*) STUB(uwu_unpatchify)
int uwu_add_pos(void*, const float*, int, int, int, int, void*) STUB(uwu_add_pos)
size_t uwu_dit_workspace_bytes(const uwu_dit_desc*) { return 0; }
int uwu_dit_forward(const uwu_dit_desc*, const float*, const float*, const float*, float*, void*) STUB(uwu_dit_forward)
int uwu_dit_backward(const uwu_dit_desc*, const float*, void*) STUB(uwu_dit_backward)
}
