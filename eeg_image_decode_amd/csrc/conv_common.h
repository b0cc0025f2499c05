// Shared between csrc/conv.hip (temporal conv + pool) and csrc/sconv.hip (BatchNorm1 -> ELU -> spatial conv): shapes of the tsconv stack
// (Retrieval/ATMS_retrieval.py:102-106) and the per-channel BatchNorm view both files use.
#pragma once
#include "eeg_common.h"

namespace eeg {

constexpr int SC_C = 40;      // channels in and out
constexpr int SC_W = 36;      // positions per row
constexpr int SC_OP = 48;     // out channels / positions padded to 3 MFMA tiles
constexpr int SCX_RS = 144;   // bytes per row of a dy2^T plane ([48 w][64 o] bf16 + 16 pad: the 16 rows of a fragment read land 4 banks apart)

struct bn_affine {            // per-channel BatchNorm as y -> xhat -> u: xhat = (y - mean) * rstd ; u = gamma * xhat + beta
    const float *mean, *rstd, *gamma, *beta;
};

}  // namespace eeg
