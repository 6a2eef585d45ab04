// The per-point callback of an observation-tree model: StarModel.lnpost(p) one point at a time, as emcee / MultiNest drive a
// generic StarModel (isochrones/starmodel.py:538-542, 797, 952).  As for BasicStarModel (iso_fast_mailbox.hip) one wave stays
// resident and polls a request in pinned, device-mapped host memory - here it is k_lnpost_tree_fast ITSELF, launched with a
// mailbox instead of a batch (no second set of instantiations): a tree has up to 24 parameters, so the request is four
// 64-byte lines (sequence word + parameters) that lanes 0-31 read in one instruction; the sequence word carries a 32-bit
// checksum of the parameter words, and a request whose lines did not arrive together is polled again.
// (shared by iso_hip.hip - host side - and iso_fast_tree.hip - the kernel)
#pragma once

namespace iso {

struct IsoTreeBox {
    unsigned long long req[32];       // [0] sequence word (checksum << 32 | counter << 16 | parts << 8), [1 ..] the parameters
    unsigned long long done[8];       // [0] sequence word of the last finished request, [1..3] lnpost, lnprior, lnlike
    unsigned long long ctl[8];        // [0] state (0 none, 1 running, 2 exited; the device writes 2), [1] quit
};

// defined in iso_fast_tree.hip: start the tree model's resident wave (false: no kernel for the shape)
bool launch_tree_mailbox(int nb, int n_leaves, const FastArgs& A, const DevTree* T, IsoTreeBox* d_box, unsigned long long idle_ticks,
                         unsigned long long life_ticks, hipStream_t s);

}  // namespace iso
