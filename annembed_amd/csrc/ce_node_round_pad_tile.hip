// ce_node_round_pad_tile.hip -- the zero-padded ce_round_node_kernel with tile negatives
#include "ce_node_round.h"

namespace ae {
void launch_round_node_padded_tile(ae_entropy_optim* o, const NodeArgs& a, uint64_t nodes) {
    const uint32_t d = o->dev.dim;
    if (d <= 8) launch_round_node_dim<8, true, true>(o, a, nodes);
    else if (d <= 16) launch_round_node_dim<16, true, true>(o, a, nodes);
    else if (d <= 32) launch_round_node_dim<32, true, true>(o, a, nodes);
    else fail(AE_ERR_INVALID_ARG, "launch_round_node_padded_tile: asked_dim > 32");
}
}  // namespace ae
