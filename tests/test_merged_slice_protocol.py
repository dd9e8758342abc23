"""A host model of the merged-slice protocol of annembed_amd/csrc/ce_slice_kernels.h (sl_slice_kernel): every class of a time slice in ONE
launch, the order between two events that share a node kept through a per-node word {classes that hold the node | classes through with
it}.  The kernel itself is tested on the GPU (tests/test_gpu_configs.py); this model pins the ARGUMENT the kernel rests on, on random
slices:
  * whatever the interleaving, a node's events run in the order of their classes (what one launch per class gives by construction);
  * workgroups dispatched in index order onto a bounded number of slots never deadlock: an event only waits for a workgroup that was
    dispatched before its own;
  * the words clean themselves: after the slice every word is zero again (the node's last event wipes it);
  * a chain through a shared target stands in the node's order as ONE event (its head waits, its last member announces it);
  * the CLASS WINDOW (round 6: a workgroup of class position q begins once every workgroup of position q - window is through -- what
    bounds the age of the negatives' rows) adds no deadlock under the same in-order dispatch, whatever the slots, and holds what it says.
No GPU, no library: pure Python."""
import numpy as np
import pytest


def random_slice(rng, n_nodes, n_events, classes):
    """events (source, target, class) of one slice whose classes are forests of in-stars: inside a class no node is the source of two
    events and none is source and target (slice_color_edges); any number may share their target (a chain)."""
    ev = []
    src_used = [set() for _ in range(classes)]
    tgt_used = [set() for _ in range(classes)]
    tries = 0
    while len(ev) < n_events and tries < 50 * n_events:
        tries += 1
        i, j, q = int(rng.integers(n_nodes)), int(rng.integers(n_nodes)), int(rng.integers(classes))
        if i == j or i in src_used[q] or i in tgt_used[q] or j in src_used[q]:
            continue
        src_used[q].add(i)
        tgt_used[q].add(j)
        ev.append((i, j, q))
    # the array order of the kernel: by class, inside a class by target (chains adjacent)
    ev.sort(key=lambda e: (e[2], e[1]))
    return ev


def run_merged(ev, n_nodes, classes, slots, wg, rng, window=0):
    """Executes the slice as the kernel does.  Returns the per-node execution log [(class, event index)].
    window > 0: a workgroup of class q begins only when the workgroups of class q - window are all through (one counter per class)."""
    cls_word = np.zeros(n_nodes, np.int64)     # low half of the word: classes with an event on the node
    done_word = np.zeros(n_nodes, np.int64)    # high half: classes through with the node
    for i, j, q in ev:                         # the marks (the slice before, or sl_dep_mark_kernel)
        cls_word[i] |= 1 << q
        cls_word[j] |= 1 << q
    # chains: maximal runs of events of one class with one target
    n = len(ev)
    head = list(range(n))
    for x in range(1, n):
        if ev[x][2] == ev[x - 1][2] and ev[x][1] == ev[x - 1][1]:
            head[x] = head[x - 1]
    last_of_chain = [x + 1 == n or head[x + 1] != head[x] for x in range(n)]
    # workgroups: runs of `wg` consecutive events of ONE class, in array order
    groups, start = [], 0
    for x in range(1, n + 1):
        if x == n or ev[x][2] != ev[start][2] or x - start == wg:
            groups.append(list(range(start, x)))
            start = x
    group_class = [ev[g[0]][2] for g in groups]
    groups_of_class = np.bincount(group_class, minlength=classes)
    class_done = np.zeros(classes, np.int64)       # SliceRunArgs::class_done
    begun = [False] * len(groups)
    log = [[] for _ in range(n_nodes)]
    attracted = [False] * n
    finished = [False] * n
    resident, next_group, steps = [], 0, 0
    while next_group < len(groups) or resident:
        while len(resident) < slots and next_group < len(groups):   # dispatch in index order
            resident.append(next_group)
            next_group += 1
        progress = False
        order = list(resident)
        rng.shuffle(order)                                           # any interleaving of the resident workgroups
        for g in order:
            if not begun[g]:
                qg = group_class[g]
                if window and qg >= window and class_done[qg - window] < groups_of_class[qg - window]:
                    continue                                         # polls its window's counter (holds its slot meanwhile)
                if window and qg >= window:                          # what the window says: that class is through, every event of it
                    assert all(finished[x] for x in range(n) if ev[x][2] == qg - window)
                begun[g] = True
            lanes = list(groups[g])
            rng.shuffle(lanes)                                       # lane by lane: any order inside a wave
            for x in lanes:
                i, j, q = ev[x]
                below = (1 << q) - 1
                if not attracted[x]:
                    need_i = cls_word[i] & below
                    need_j = (cls_word[j] & below) if head[x] == x else 0          # a chain's head stands for the chain
                    follows_ok = head[x] == x or attracted[x - 1]                  # the lane before it in the chain
                    if (done_word[i] & need_i) == need_i and (done_word[j] & need_j) == need_j and follows_ok:
                        log[i].append((q, x))
                        log[j].append((q, x))
                        attracted[x] = True
                        progress = True
                        if last_of_chain[x]:                                       # the target's row is final for this class
                            if cls_word[j] >> (q + 1):
                                done_word[j] |= 1 << q
                            else:
                                cls_word[j] = 0; done_word[j] = 0                  # the node's last event of the slice wipes the word
                elif not finished[x]:                                              # repulsions, the source's store
                    if cls_word[i] >> (q + 1):
                        done_word[i] |= 1 << q
                    else:
                        cls_word[i] = 0; done_word[i] = 0
                    finished[x] = True
                    progress = True
        for g in resident:
            if all(finished[x] for x in groups[g]):
                class_done[group_class[g]] += 1
        resident = [g for g in resident if not all(finished[x] for x in groups[g])]
        steps += 1
        assert progress, "no resident lane could run: deadlock (step %d, %d groups resident, next %d of %d)" % (steps, len(resident), next_group, len(groups))
    assert not cls_word.any() and not done_word.any(), "words left dirty after the slice"
    return log


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("slots", [1, 3, 64])
def test_node_order_is_the_class_order_and_nothing_hangs(seed, slots):
    rng = np.random.default_rng(seed)
    classes = int(rng.integers(2, 16))
    n_nodes = int(rng.integers(20, 200))
    ev = random_slice(rng, n_nodes, int(rng.integers(50, 400)), classes)
    log = run_merged(ev, n_nodes, classes, slots=slots, wg=int(rng.choice([2, 4, 8])), rng=rng)
    seen = 0
    for v in range(n_nodes):
        qs = [q for q, _ in log[v]]
        assert qs == sorted(qs), (v, log[v])           # a node's events in the order of their classes ...
        for a in range(1, len(log[v])):                # ... and inside a class (a chain through the target) in array order
            if log[v][a][0] == log[v][a - 1][0]:
                assert log[v][a][1] > log[v][a - 1][1], (v, log[v])
        seen += len(qs)
    assert seen == 2 * len(ev)                         # every event ran once, on both its rows


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("slots,window", [(1, 1), (2, 1), (3, 2), (64, 3), (5, 8)])
def test_class_window_neither_hangs_nor_reorders(seed, slots, window):
    rng = np.random.default_rng(100 + seed)
    classes = int(rng.integers(3, 16))
    n_nodes = int(rng.integers(20, 200))
    ev = random_slice(rng, n_nodes, int(rng.integers(50, 400)), classes)
    log = run_merged(ev, n_nodes, classes, slots=slots, wg=int(rng.choice([2, 4, 8])), rng=rng, window=window)
    for v in range(n_nodes):
        qs = [q for q, _ in log[v]]
        assert qs == sorted(qs), (v, log[v])
    assert sum(len(l) for l in log) == 2 * len(ev)


def test_a_hub_chain_counts_as_one_event_of_its_node():
    # node 0 is the target of five events of class 1 (a chain), the source of one event of class 0 and of one of class 2
    ev = [(0, 9, 0)] + [(s, 0, 1) for s in (3, 4, 5, 6, 7)] + [(0, 8, 2)]
    log = run_merged(ev, 10, 3, slots=2, wg=2, rng=np.random.default_rng(1))
    assert [q for q, _ in log[0]] == [0, 1, 1, 1, 1, 1, 2]
    assert [x for _, x in log[0]][1:6] == [1, 2, 3, 4, 5]
