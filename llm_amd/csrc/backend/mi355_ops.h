// mi355_ops.h -- the non-matmul ggml ops the backend serves so a Llama/Mixtral graph has no CPU
// splits.  Implemented in ops.cpp on top of the C ABI kernels.
#pragma once
#include "ggml.h"
#include <hip/hip_runtime.h>

bool mi355_ops_supported(const ggml_tensor * op);
bool mi355_ops_forward(ggml_tensor * node, hipStream_t stream);
