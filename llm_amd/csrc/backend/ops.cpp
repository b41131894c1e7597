// ops.cpp -- non-matmul ops of the backend (see mi355_ops.h).
#include "mi355_ops.h"

bool mi355_ops_supported(const ggml_tensor *) { return false; }
bool mi355_ops_forward(ggml_tensor *, hipStream_t) { return false; }
