// ce_node_round_exact.hip -- ce_round_node_kernel for the dimensions with a vector-load instantiation (gathered negatives:
// hubness sampler, small graphs, A/B)
#include "ce_node_round.h"

namespace ae {
void launch_round_node_exact(ae_entropy_optim* o, const NodeArgs& a, uint64_t nodes) {
    switch (o->dev.dim) {
        case 2: launch_round_node_dim<2, false, false>(o, a, nodes); break;
        case 3: launch_round_node_dim<3, false, false>(o, a, nodes); break;
        case 4: launch_round_node_dim<4, false, false>(o, a, nodes); break;
        case 8: launch_round_node_dim<8, false, false>(o, a, nodes); break;
        case 16: launch_round_node_dim<16, false, false>(o, a, nodes); break;
        default: fail(AE_ERR_INVALID_ARG, "launch_round_node_exact: dimension without an exact instantiation");
    }
}
}  // namespace ae
