// ce_node_round_pad.hip -- ce_round_node_kernel for any other asked_dim <= 32, rows zero-padded in registers (gathered negatives)
#include "ce_node_round.h"

namespace ae {
void launch_round_node_padded(ae_entropy_optim* o, const NodeArgs& a, uint64_t nodes) {
    const uint32_t d = o->dev.dim;
    if (d <= 8) launch_round_node_dim<8, true, false>(o, a, nodes);
    else if (d <= 16) launch_round_node_dim<16, true, false>(o, a, nodes);
    else if (d <= 32) launch_round_node_dim<32, true, false>(o, a, nodes);
    else fail(AE_ERR_INVALID_ARG, "launch_round_node_padded: asked_dim > 32");
}
}  // namespace ae
