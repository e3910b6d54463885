#include "../../torch/serialize/tensor.h"
