// Temporary: entries declared in include/uwu_hip.h that are not implemented yet return UWU_ENOTIMPL.
#include <stddef.h>
#include <stdint.h>
#include "../../include/uwu_hip.h"
void uwu_set_error(const char* fmt, ...);
#define STUB(name) { uwu_set_error(#name ": not implemented"); return UWU_ENOTIMPL; }
extern "C" {
size_t uwu_dit_workspace_bytes(const uwu_dit_desc*) { return 0; }
int uwu_dit_forward(const uwu_dit_desc*, const float*, const float*, const float*, float*, void*) STUB(uwu_dit_forward)
int uwu_dit_backward(const uwu_dit_desc*, const float*, void*) STUB(uwu_dit_backward)
}
