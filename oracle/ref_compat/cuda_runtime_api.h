#include "cuda_runtime.h"
