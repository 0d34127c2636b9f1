// k_train.hip - the training step: backward / optimiser kernels (train.hpp) and their host side + C ABI (train_host.hpp).
#include "host.hpp"
#include "train.hpp"
#include "fused_bwd.hpp"

using namespace mpdx;

#include "train_host.hpp"
