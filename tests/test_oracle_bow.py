"""Oracle of the bag-of-words side of the loop detector (oracle/bow.c: DBoW2's vocabulary tree, TF-IDF BowVector, L1 score,
direct index) against numpy restatements of its parts written from the formulas, and the properties that make it DBoW2's:
node numbering, word weights, scores in [0, 1] with 1 for identical images, the file format round trip."""
import numpy as np
import pytest

from bow_fixtures import noisy_descriptor_images
from oracle import orc
from ros_stereo_slam_amd import vocabulary


def _ham(a, b):
    return int(sum(bin(int(x) ^ int(y)).count("1") for x, y in zip(a, b)))


@pytest.fixture(scope="module")
def trained():
    imgs = noisy_descriptor_images()
    return imgs, orc.Vocabulary.train(imgs, k=9, L=4, seed=3)


def test_tree_structure_and_dbow2_numbering(trained):
    imgs, v = trained
    a = v.arrays()
    n = v.n_nodes
    assert a["parent"][0] == -1 and (a["parent"][1:] < np.arange(1, n)).all()        # parents come first
    for p in range(n):                                                                 # children consecutive
        ch = np.nonzero(a["parent"] == p)[0]
        assert len(ch) == a["n_children"][p] and len(ch) <= 9
        if len(ch):
            assert a["first_child"][p] == ch[0] and (np.diff(ch) == 1).all()
    # creation order: a node's children are numbered before any grandchild, then depth first -- node 1..9 are the root's
    assert (a["parent"][1:1 + a["n_children"][0]] == 0).all()
    assert a["parent"][1 + a["n_children"][0]] == 1                                    # then the children of node 1
    leaves = np.nonzero(a["n_children"][1:] == 0)[0] + 1
    assert np.array_equal(a["word_id"][leaves], np.arange(len(leaves))) and v.n_words == len(leaves)
    assert (a["word_id"][a["n_children"] > 0] == -1).all()
    depth = np.zeros(n, int)
    for i in range(1, n):
        depth[i] = depth[a["parent"][i]] + 1
    assert depth.max() <= 4


def test_one_node_clustering_against_numpy(trained):
    """HKmeansStep of one node: the final state is a fixed point of (majority vote, nearest centre with first-minimum
    ties), every cluster keeps at least its seed... and the seeding follows the stated draws."""
    imgs, _ = trained
    D = np.concatenate(imgs[:6])
    idx = np.arange(len(D), dtype=np.int32)
    cen, assoc, steps = orc.voc_cluster(D, idx, 9, seed=5, key=1)
    assert len(cen) == 9 and steps >= 1
    bits = ((D[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(len(D), 256)
    for c in range(9):
        m = bits[assoc == c]
        assert len(m) > 0
        want = (m.sum(0) >= (len(m) + 1) // 2)                                       # FORB::meanValue
        got = ((cen[c][:, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(256).astype(bool)
        assert np.array_equal(got, want), c
    dist = np.array([[_ham(d, c) for c in cen] for d in D[:400]])
    assert np.array_equal(dist.argmin(1), assoc[:400])                                # argmin = first minimum
    # at most k descriptors: one cluster each, in order
    cen2, assoc2, steps2 = orc.voc_cluster(D, idx[:7], 9, seed=5, key=1)
    assert len(cen2) == 7 and np.array_equal(assoc2, np.arange(7)) and np.array_equal(cen2, D[:7]) and steps2 == 0
    # deterministic, and the key matters
    cen3, assoc3, _ = orc.voc_cluster(D, idx, 9, seed=5, key=1)
    assert np.array_equal(cen, cen3) and np.array_equal(assoc, assoc3)
    cen4, _, _ = orc.voc_cluster(D, idx, 9, seed=5, key=18)
    assert not np.array_equal(cen, cen4)


def test_transform_weights_and_bow_vector_against_numpy(trained):
    imgs, v = trained
    a = v.arrays()
    # descend by hand
    d = imgs[3][17]
    cur, path = 0, []
    while a["n_children"][cur] > 0:
        ch = np.arange(a["first_child"][cur], a["first_child"][cur] + a["n_children"][cur])
        cur = int(ch[np.argmin([_ham(d, a["desc"][c]) for c in ch])])
        path.append(cur)
    word, weight, node = v.transform(imgs[3], levelsup=2)
    assert word[17] == a["word_id"][cur] and weight[17] == a["weight"][cur]
    want_node = path[v.L - 2 - 1] if len(path) >= v.L - 2 else 0
    assert node[17] == want_node
    # idf weights: log(N / Ni) with Ni = images containing the word
    words_per_image = [set(v.transform(im)[0].tolist()) for im in imgs]
    leaf_of_word = {int(a["word_id"][i]): i for i in range(v.n_nodes) if a["word_id"][i] >= 0}
    for w in list(leaf_of_word)[:200]:
        ni = sum(w in s for s in words_per_image)
        want = np.log(len(imgs) / ni) if ni else 0.0
        assert abs(a["weight"][leaf_of_word[w]] - want) <= 1e-15 * max(1.0, want)
    # BowVector: tf-idf accumulated per word, L1-normalised, ascending words, zero-weight words dropped
    bw, bv, nd = v.bow(imgs[3], 2)
    acc = {}
    for w_, wt in zip(word, weight):
        if wt > 0:
            acc[int(w_)] = acc.get(int(w_), 0.0) + wt
    ws = sorted(acc)
    vals = np.array([acc[w_] for w_ in ws])
    assert np.array_equal(bw, ws) and np.allclose(bv, vals / vals.sum(), rtol=1e-14, atol=0)
    assert abs(bv.sum() - 1.0) < 1e-12 and (np.diff(bw) > 0).all()
    assert np.array_equal(nd, np.where(weight > 0, node, -1))


def test_l1_score_properties_and_database_query(trained):
    imgs, v = trained
    bows = [v.bow(im, 2) for im in imgs]
    s_self, c_self = orc.bow_l1_sum(bows[0][0], bows[0][1], bows[0][0], bows[0][1])
    assert abs(-s_self / 2 - 1.0) < 1e-12 and c_self == len(bows[0][0])              # an image scores 1 against itself
    for i, j in ((0, 1), (2, 9), (5, 30)):
        wi, vi, _ = bows[i]
        wj, vj, _ = bows[j]
        s, c = orc.bow_l1_sum(wi, vi, wj, vj)
        common = sorted(set(wi.tolist()) & set(wj.tolist()))
        di, dj = dict(zip(wi.tolist(), vi)), dict(zip(wj.tolist(), vj))
        want = sum(abs(di[w] - dj[w]) - abs(di[w]) - abs(dj[w]) for w in common)
        assert c == len(common) and abs(s - want) < 1e-14
        full = 1.0 - 0.5 * sum(abs(di.get(w, 0.0) - dj.get(w, 0.0)) for w in set(di) | set(dj))   # 1 - ||v - w||_1 / 2
        assert abs(-s / 2 - full) < 1e-12 and 0.0 <= -s / 2 <= 1.0
    stride = max(len(b[0]) for b in bows)
    dbw, dbv, dbn = np.zeros((len(bows), stride), np.int32), np.zeros((len(bows), stride)), np.zeros(len(bows), np.int32)
    for e, (w, val, _) in enumerate(bows):
        dbn[e] = len(w)
        dbw[e, :len(w)], dbv[e, :len(w)] = w, val
    sums, common = orc.bow_query(bows[7][0], bows[7][1], dbw, dbv, dbn)
    assert np.argmin(sums) == 7
    for e in (0, 3, 12):
        assert sums[e] == orc.bow_l1_sum(bows[7][0], bows[7][1], bows[e][0], bows[e][1])[0]


def test_direct_index_matching_against_a_python_restatement(trained):
    imgs, v = trained
    rng = np.random.default_rng(4)
    A = imgs[0]
    noise = ((rng.random((len(A), 8, 32)) < 0.02) * (1 << np.arange(32, dtype=np.uint64))).sum(axis=2).astype(np.uint32)
    B = np.ascontiguousarray((A ^ noise)[rng.permutation(len(A))])                    # the same image, perturbed, shuffled
    _, _, na = v.bow(A, 2)
    _, _, nb = v.bow(B, 2)
    io, ic = orc.di_matches(A, na, B, nb, 0.6)
    # restated from TemplatedLoopDetector.h:1005-1054, 1255-1316 with python containers
    want_o, want_c = [], []
    for node in sorted(set(na[na >= 0].tolist()) & set(nb[nb >= 0].tolist())):
        ia, ib = np.nonzero(na == node)[0], np.nonzero(nb == node)[0]
        mo, mc = [], []
        for i in ia:
            d = [_ham(A[i], B[j]) for j in ib]
            o = np.argsort(d, kind="stable")
            b1 = d[o[0]]
            b2 = d[o[1]] if len(o) > 1 else 1e9
            if b2 == 0 and b1 == 0:
                continue
            if b1 / b2 <= 0.6:
                j = int(ib[o[0]])
                if j not in mc:
                    mc.append(j)
                    mo.append(int(i))
                elif b1 < _ham(A[mo[mc.index(j)]], B[j]):
                    mo[mc.index(j)] = int(i)
        want_o += mo
        want_c += mc
    assert io.tolist() == want_o and ic.tolist() == want_c and len(io) > 50


def test_dbow2_file_round_trip(trained, tmp_path):
    imgs, v = trained
    a = v.arrays()
    for name in ("voc.yml", "voc.yml.gz"):
        path = tmp_path / name
        vocabulary.save_dbow2(path, v.k, v.L, a["parent"], a["desc"], a["weight"], a["word_id"])
        got = vocabulary.load_dbow2(path)
        assert got["k"] == 9 and got["L"] == 4 and got["scoring"] == 0 and got["weighting"] == 0 and not got["renumbered"]
        assert np.array_equal(got["parent"], a["parent"]) and np.array_equal(got["desc"], a["desc"])
        assert np.array_equal(got["weight"], a["weight"]) and np.array_equal(got["word_id"], a["word_id"])
    v2 = orc.Vocabulary.from_arrays(got["k"], got["L"], got["parent"], got["desc"], got["weight"])
    assert v2.n_words == v.n_words
    for im in imgs[:3]:
        for x, y in zip(v.bow(im, 2), v2.bow(im, 2)):
            assert np.array_equal(x, y)
    text = open(tmp_path / "voc.yml").read()
    assert text.startswith("%YAML:1.0") and "nodeId:1, parentId:0" in text and "wordId:0, nodeId:" in text
    first = text.split('descriptor:"')[1].split('"')[0].split()
    assert len(first) == 32 and [int(b) for b in first] == a["desc"][1].view(np.uint8).tolist()
