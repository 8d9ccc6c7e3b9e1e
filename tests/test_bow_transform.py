"""DBoW2's vocabulary transform (the step after extraction for every keyframe, reference
src/Frame.cc:513-522) with the vocabulary as plain arrays.

DBoW2 and its vocabulary file are not vendored by the reference, so there is no fixture to pin
this against: parity unpinned.  The oracle restates the published traversal; here it is checked
against an independent numpy statement of the definition on synthetic trees (k = 10, L = 3, plus an
irregular tree with ties, wide nodes and leaves at several depths), and the HIP path against the
oracle bit for bit.
"""
import numpy as np
import pytest

from slam_amd import synth


def make_tree(k, levels, seed, irregular=False):
    """A vocabulary tree as arrays.  Children are perturbed copies of their parent, so that
    descents are decisive; the irregular tree adds duplicate children (ties), one node with 23
    children, and leaves above the last level."""
    r = np.random.default_rng(seed)
    desc = [np.zeros(32, np.uint8)]
    kids = [[]]
    depth = [0]
    frontier = [0]
    for lvl in range(1, levels + 1):
        nxt = []
        for p in frontier:
            if irregular and lvl > 1 and r.uniform() < 0.15:
                continue                                   # this node stays a leaf
            nk = 23 if (irregular and lvl == 2 and p == frontier[0]) else (int(r.integers(2, k + 1)) if irregular else k)
            for c in range(nk):
                d = desc[p].copy() if lvl > 1 else r.integers(0, 256, 32).astype(np.uint8)
                flips = r.integers(0, 256, 40 // lvl)
                for b in flips:
                    d[b >> 3] ^= np.uint8(1 << (b & 7))
                if irregular and c > 0 and r.uniform() < 0.1:
                    d = desc[kids[p][-1]].copy()            # same descriptor as the previous sibling: a tie
                desc.append(d)
                kids.append([])
                depth.append(lvl)
                kids[p].append(len(desc) - 1)
                nxt.append(len(desc) - 1)
        frontier = nxt
    n = len(desc)
    child_start = np.zeros(n + 1, np.int32)
    children = []
    for i in range(n):
        children += kids[i]
        child_start[i + 1] = len(children)
    word_id = np.full(n, -1, np.int32)
    weight = np.zeros(n, np.float64)
    leaves = [i for i in range(n) if not kids[i]]
    for w, i in enumerate(leaves):
        word_id[i] = w
        weight[i] = float(r.uniform(0.2, 9.0))              # an idf-like weight
    weight[leaves[3]] = 0.0                                 # a stopped word
    return dict(levels=levels, node_desc=np.array(desc, np.uint8), child_start=child_start,
                children=np.array(children, np.int32), word_id=word_id, weight=weight), leaves


def descend_numpy(voc, f, levelsup):
    """The definition, independently of the oracle: closest child, first on ties, until a leaf."""
    node, level, nid = 0, 0, 0
    nid_level = voc["levels"] - levelsup
    while voc["child_start"][node + 1] > voc["child_start"][node]:
        ch = voc["children"][voc["child_start"][node]:voc["child_start"][node + 1]]
        d = np.unpackbits(voc["node_desc"][ch] ^ f[None, :], axis=1).sum(axis=1)
        node = int(ch[int(np.argmin(d))])                  # argmin returns the first minimum
        level += 1
        if level == nid_level:
            nid = node
    return int(voc["word_id"][node]), float(voc["weight"][node]), nid


def features_for(voc, leaves, n, seed):
    r = np.random.default_rng(seed)
    pick = r.integers(0, len(leaves), n)
    f = synth.perturb_descriptors(voc["node_desc"][np.array(leaves)[pick]], seed, 12)
    f[::7] = r.integers(0, 256, (len(f[::7]), 32)).astype(np.uint8)     # some unrelated features
    return f


@pytest.mark.parametrize("k,levels,irregular,levelsup", [(10, 3, False, 1), (10, 3, False, 4), (6, 4, True, 2), (4, 5, True, 4)])
def test_oracle_transform_equals_the_definition(oracle, k, levels, irregular, levelsup):
    voc, leaves = make_tree(k, levels, 5 + k, irregular)
    f = features_for(voc, leaves, 300, 9)
    word, weight, node = oracle.bow_transform(voc, f, levelsup)
    for i in range(len(f)):
        w, wt, nid = descend_numpy(voc, f[i], levelsup)
        assert (word[i], weight[i], node[i]) == (w, wt, nid), i
    if not irregular:
        # a feature a few bits away from a leaf reaches that leaf (children differ by far more bits)
        g = synth.perturb_descriptors(voc["node_desc"][np.array(leaves[:50])], 3, 4)
        word, _, _ = oracle.bow_transform(voc, g, levelsup)
        assert np.array_equal(word, voc["word_id"][np.array(leaves[:50])])


def test_oracle_bow_vector_known_answer(oracle):
    word = np.array([5, 2, 5, 9, 2, 5, 7], np.int32)
    weight = np.array([1.5, 0.25, 1.5, 0.0, 0.25, 1.5, 2.0], np.float64)   # word 9 is stopped
    ids, vals = oracle.bow_vector(word, weight)
    assert list(ids) == [2, 5, 7]
    s = (0.25 + 0.25) + ((1.5 + 1.5) + 1.5) + 2.0
    assert np.array_equal(vals, np.array([0.5 / s, 4.5 / s, 2.0 / s]))


@pytest.mark.gpu
@pytest.mark.parametrize("k,levels,irregular,levelsup", [(10, 3, False, 1), (10, 3, False, 4), (10, 6, False, 4), (6, 4, True, 2),
                                                         (4, 5, True, 1)])
def test_hip_transform_equals_oracle(oracle, k, levels, irregular, levelsup):
    from slam_amd.matcher import ORBmatcher
    if levels == 6:
        voc, leaves = make_tree(4, 6, 77, False)           # 5461 nodes: the depth of the ORB vocabulary
    else:
        voc, leaves = make_tree(k, levels, 5 + k, irregular)
    f = features_for(voc, leaves, 2500, 11)
    m = ORBmatcher(0.6, True)
    try:
        ids, vals, fv = m.BowTransform(voc, f, levelsup)
        e_ids, e_vals, e_fv = m.BowTransform(voc, f[:0], levelsup)
        assert len(e_ids) == 0 and len(e_vals) == 0 and e_fv == {}
    finally:
        m.close()
    word, weight, node = oracle.bow_transform(voc, f, levelsup)
    oids, ovals = oracle.bow_vector(word, weight)
    assert np.array_equal(ids, oids)
    assert np.array_equal(vals.view(np.uint64), ovals.view(np.uint64))
    want = {}
    for i in range(len(f)):
        if weight[i] > 0:
            want.setdefault(int(node[i]), []).append(i)
    assert fv == want
    assert len(fv) > 1 if voc["levels"] - levelsup > 0 else list(fv) == [0]
