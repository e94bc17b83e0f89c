"""CPU-only checks: the C-ABI library loads and exports every symbol include/icz.h declares (no compute calls
without a GPU), host-side logic (packing order, sharding, vocabulary, CIDEr-D hashing / cooking) and the synthetic
data generators."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "icz.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(icz_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from simpleimagecaptionzoo_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "libicz.so missing: run __graft_entry__.build()"
    L = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 20
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing
    L.icz_version.restype = ctypes.c_char_p
    assert b"gfx950" in L.icz_version()
    _lib.lib()      # the binding table itself resolves


def test_argument_validation_without_gpu():
    """Calls that must fail before touching the device: status + message through icz_last_error."""
    from simpleimagecaptionzoo_amd._lib import ButdDims, lib
    L = lib()
    h = ctypes.c_void_p()
    bad = ButdDims(100, 2048, 1024, 1024, 1024, 10102, 64, 20)      # R > 64
    assert L.icz_butd_create(ctypes.byref(bad), ctypes.byref(h)) == -1
    assert b"R=100" in L.icz_last_error()
    bad = ButdDims(36, 2047, 1024, 1024, 1024, 10102, 64, 20)       # D not a multiple of 4
    assert L.icz_butd_create(ctypes.byref(bad), ctypes.byref(h)) == -1
    assert L.icz_butd_create(None, ctypes.byref(h)) == -1
    assert L.icz_ciderd_create(None, None, 8, 0.0, None, ctypes.byref(h)) == -1
    assert L.icz_adam_clamp_step(None, None, None, None, 0, 0.0, 0.0, 1, None) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from simpleimagecaptionzoo_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.IczError, match="no fallback"):
        _lib.lib()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "simpleimagecaptionzoo_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, fn)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, fn


def test_vocab_and_param_tree_match_reference_layout():
    from simpleimagecaptionzoo_amd._lib import BUTD_PARAM_KEYS
    from simpleimagecaptionzoo_amd.captioner import _DecoderParams
    from simpleimagecaptionzoo_amd.vocab import synthetic_vocab
    v = synthetic_vocab(50)
    assert [v.ix2word[i] for i in range(4)] == ["<pad>", "<sta>", "<end>", "<unk>"] and len(v) == 50
    assert v("not-a-word") == 3 and v("w0") == 4
    d = _DecoderParams(atten_dim=8, embed_dim=12, hidden_dim=16, vocab_size=50, enc_dim=32)
    sd = d.state_dict()
    assert sorted(sd.keys()) == sorted(BUTD_PARAM_KEYS)
    assert tuple(sd["TD_atten.weight_ih"].shape) == (64, 16 + 32 + 12)
    assert tuple(sd["language_model.weight_ih"].shape) == (64, 32 + 16)
    assert tuple(sd["atten.affine.weight_g"].shape) == (1, 1)
    assert tuple(sd["predict.weight_v"].shape) == (50, 16)
    # weight_norm initial state: g = ||v|| per row, predict.bias = 0 (BUTD_Model.py:87-90)
    np.testing.assert_allclose(sd["predict.weight_g"].squeeze(1).numpy(), sd["predict.weight_v"].norm(dim=1).numpy(), rtol=1e-6)
    assert float(sd["predict.bias"].abs().max()) == 0.0
    assert float(sd["embed.0.weight"].abs().max()) <= 0.1 and float(sd["predict.weight_v"].abs().max()) <= 0.1


def test_shard_range_partitions_exactly():
    from simpleimagecaptionzoo_amd.dist import shard_range
    for n in (1, 7, 64, 65, 128):
        for w in (1, 2, 3, 8):
            cuts = [shard_range(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_ciderd_hash_twin_and_philox_reference_vectors():
    """numpy twin of the device n-gram hash (open addressing needs bit-identical hashing on both sides)."""
    from simpleimagecaptionzoo_amd.ciderd import _hash_keys

    def ref(a, b, c, d):
        h = 2166136261
        for x in (a, b, c, d):
            h = ((h ^ (x & 0xFFFFFFFF)) * 16777619) & 0xFFFFFFFF
        return h ^ (h >> 15)
    keys = np.array([[5, -1, -1, -1], [1, 2, 3, 4], [10101, 7, -1, -1], [0, 0, 0, 0]], dtype=np.int32)
    assert [int(x) for x in _hash_keys(keys)] == [ref(*[int(v) for v in k]) for k in keys]


def test_synthetic_generators_are_deterministic():
    from simpleimagecaptionzoo_amd.synth import document_frequency, random_butd_params, synthetic_references
    words = ["<pad>", "<sta>", "<end>", "<unk>"] + ["w%d" % i for i in range(96)]
    a, b = synthetic_references(5, words, seed=3), synthetic_references(5, words, seed=3)
    assert a == b and all(len(v) == 5 for v in a.values())
    df = document_frequency(a)
    assert df["ref_len"] == 5 and max(df["document_frequency"].values()) <= 5
    p1 = random_butd_params(36, 32, 16, 16, 16, 50, "cpu", seed=1)
    p2 = random_butd_params(36, 32, 16, 16, 16, 50, "cpu", seed=1)
    assert all(torch.equal(p1[k], p2[k]) for k in p1)


def test_ptb_lite_tokenizer_and_annotation_loader(tmp_path):
    import json
    from simpleimagecaptionzoo_amd.coco_eval import load_annotations, ptb_lite_tokenize, tokenize
    assert ptb_lite_tokenize("A man riding a wave on top of a surfboard.") == "a man riding a wave on top of a surfboard"
    assert ptb_lite_tokenize("Two dogs, one black; one white -- can't sit!") == "two dogs one black one white ca n't sit"
    assert ptb_lite_tokenize("The dog's toy isn't here... it's the cats' toy") == "the dog 's toy is n't here it 's the cats toy"
    assert ptb_lite_tokenize('He said "hello" (twice) at 3 o\'clock.') == "he said hello -lrb- twice -rrb- at 3 o'clock"
    assert ptb_lite_tokenize("") == "" and ptb_lite_tokenize(" . ") == ""
    # Penn-Treebank conventions beyond plain captions (UNPINNED against the Stanford jar, see coco_eval.py): each is the
    # documented PTB3 behaviour, lower-cased, punctuation tokens of ptbtokenizer.py:24-25 dropped afterwards
    assert ptb_lite_tokenize("I'm gonna say we've got 1,000 kites at 3:30, OK?") == "i 'm gon na say we 've got 1,000 kites at 3:30 ok"
    assert ptb_lite_tokenize("You cannot park here; they'd tow it.") == "you can not park here they 'd tow it"
    assert ptb_lite_tokenize("A $5 pizza & 50% off") == "a $ 5 pizza & 50 % off"
    assert ptb_lite_tokenize("Mr. Smith's U.S. flag.") == "mr. smith 's u.s. flag"
    assert ptb_lite_tokenize("a well-lit, snow-covered street") == "a well-lit snow-covered street"
    assert ptb_lite_tokenize("children: two boys, a girl") == "children two boys a girl"
    ann = {"annotations": [{"image_id": 7, "caption": "A cat."}, {"image_id": 9, "caption": "Dogs!"}, {"image_id": 7, "caption": "Cat sits"}]}
    p = tmp_path / "ann.json"
    p.write_text(json.dumps(ann))
    got = load_annotations(str(p))
    assert list(got.keys()) == [7, 9] and [c["caption"] for c in got[7]] == ["A cat.", "Cat sits"]
    assert tokenize(got) == {7: ["a cat", "cat sits"], 9: ["dogs"]}


def _write_ref_feature_files(root, ids, R=36, D=64, seed=0):
    """The on-disk layout of PreProcess/Generate_coco14_bottom_up_features_data.py:56-58 that Datasets.py:54-58 reads."""
    import numpy as np
    rng = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, "fixed_bu_feat"))
    os.makedirs(os.path.join(root, "fixed_bu_bbox"))
    data = {}
    for i in ids:
        f = rng.rand(R, D).astype(np.float32)
        b = rng.rand(R, 4).astype(np.float32)
        np.savez_compressed(os.path.join(root, "fixed_bu_feat", "%s.npz" % i), feat=f)
        np.save(os.path.join(root, "fixed_bu_bbox", "%s.npy" % i), b)
        data[i] = (f, b)
    return data


def test_packed_feature_store_roundtrip(tmp_path):
    import numpy as np
    from simpleimagecaptionzoo_amd.features import PackedFeatureStore, pack_npz_dir
    ids = [391895, 522418, 184613, 318219]
    data = _write_ref_feature_files(str(tmp_path / "supp"), ids)
    prefix = pack_npz_dir(str(tmp_path / "supp"), ids, str(tmp_path / "packed"))
    store = PackedFeatureStore(prefix)
    assert len(store) == 4 and 522418 in store and 1 not in store and (store.R, store.D) == (36, 64)
    for i in ids:
        assert np.array_equal(store[i]["bu_feat"], data[i][0]) and np.array_equal(store[i]["bu_bbox"], data[i][1])
    out = np.zeros((2, 36, 64), np.float32)
    store.gather_into([184613, 391895], out)
    assert np.array_equal(out[0], data[184613][0]) and np.array_equal(out[1], data[391895][0])


def test_packed_feature_store_adaptive_roundtrip(tmp_path):
    """'adaptive' feature sets (10..100 rows per image, Datasets.py:59-61): ragged store, batches padded with zero rows."""
    import numpy as np
    from simpleimagecaptionzoo_amd.features import PackedFeatureStore, pack_npz_dir
    root = str(tmp_path / "supp")
    os.makedirs(os.path.join(root, "adaptive_bu_feat"))
    os.makedirs(os.path.join(root, "adaptive_bu_bbox"))
    rng = np.random.RandomState(3)
    data = {}
    for i, n in ((11, 10), (12, 100), (13, 37)):
        data[i] = (rng.rand(n, 32).astype(np.float32), rng.rand(n, 4).astype(np.float32))
        np.savez_compressed(os.path.join(root, "adaptive_bu_feat", "%d.npz" % i), feat=data[i][0])
        np.save(os.path.join(root, "adaptive_bu_bbox", "%d.npy" % i), data[i][1])
    store = PackedFeatureStore(pack_npz_dir(root, [11, 12, 13], str(tmp_path / "packed"), kind="adaptive"))
    assert store.ragged and len(store) == 3 and (store.R, store.D) == (100, 32) and store.counts([13, 11]) == [37, 10]
    for i in data:
        assert np.array_equal(store[i]["bu_feat"], data[i][0]) and np.array_equal(store[i]["bu_bbox"], data[i][1])
    out = np.full((2, 37, 32), 7.0, np.float32)
    store.gather_into([13, 11], out)
    assert np.array_equal(out[0], data[13][0]) and np.array_equal(out[1, :10], data[11][0]) and not out[1, 10:].any()


def test_scheduled_sampling_schedule_and_captioner_plumbing():
    """Engine.py:140-144's schedule, and the Captioner-side state that carries `ss_prob` to the device handle."""
    from simpleimagecaptionzoo_amd.scheduled import ScheduledSamplingState, scheduled_sampling_prob
    o = {"ss_start_epoch": 2, "ss_inc_every": 3, "ss_inc_prob": 0.05, "ss_max_prob": 0.25}
    assert [scheduled_sampling_prob(e, o) for e in (0, 2, 3, 4, 5, 8, 30)] == [0.0, 0.0, 0.0, 0.0, 0.05, 0.1, 0.25]
    assert scheduled_sampling_prob(9, dict(o, ss_start_epoch=-1)) == 0.0

    class Handle:
        def __init__(self):
            self.calls = []

        def set_scheduled_sampling(self, p, gate, draw):
            self.calls.append((p, gate, draw))

    st = ScheduledSamplingState()
    st._ss_init()
    h = Handle()
    st._ss_push(h, True)
    assert h.calls == []                      # a new handle is off already
    st.ss_prob = 0.25                         # what Engine.py:143 does: ignored by default, as in the reference's runs
    st._ss_push(h)
    assert h.calls == []
    st.scheduled_sampling = True              # opt in: the attribute is live
    st._ss_push(h)
    st._ss_push(h)
    assert h.calls == [(0.25, None, None)]
    h2 = Handle()
    st._ss_push(h2, True)                     # a re-created handle gets the live value again
    assert h2.calls == [(0.25, None, None)]
    st.ss_prob = 0.0
    st._ss_push(h2)
    assert h2.calls[-1] == (0.0, None, None)


def test_host_reference_cooker_matches_the_python_statement():
    """icz_ciderd_cook_host (host code of libicz: n-gram counting in the scorer's dict order, tf-idf weights, norms, bigram
    length; ciderD_scorer.py:17-32, 128-153) against the Python statement of the same in ciderd.ReferenceCooker.cook_image:
    every array bit-identical -- with out-of-vocabulary words in the references AND in the document-frequency table, empty
    and one-word references, repeated n-grams."""
    from simpleimagecaptionzoo_amd.ciderd import ReferenceCooker
    from simpleimagecaptionzoo_amd.synth import document_frequency, synthetic_references
    from simpleimagecaptionzoo_amd.vocab import synthetic_vocab
    v = synthetic_vocab(300)
    words = [v.ix2word[i] for i in range(300)]
    train = synthetic_references(200, words + ["oov%d" % i for i in range(40)], seed=0)
    df = document_frequency(train)
    ck = ReferenceCooker(df["document_frequency"], df["ref_len"], v.word2ix)
    refs = synthetic_references(24, words + ["oov%d" % i for i in range(60)], seed=3)
    refs[3] = ["", "w1", "w1 w1 w1 w1 w1", "w2 w3 w2 w3 w2 w3", "oov3 oov3 zzz", "<pad>"]
    refs[7] = ["w5 w6 w7 w8 w9 w10 w11 w12 w13 w14 w15 w16 w17 w18 w19 w20 w21 w22 w23 w24"]
    want = [ck.cook_image(refs[i]) for i in range(24)]
    got = ck.cook_images([refs[i] for i in range(24)])
    for a, b in zip(want, got):
        for x, y in zip(a, b):
            assert x.dtype == y.dtype and x.shape == y.shape and np.array_equal(x, y)
    assert any((a[1] >= 300).any() for a in want)          # private ids of out-of-vocabulary words are in play


def test_rank_batches_shards_without_touching_foreign_batches():
    """Data-parallel evaluation (engine._rank_batches): rank r gets the batches i with i % world == r -- an indexable loader
    is only asked for those (no feature I/O for the others), a loader with shard() is delegated to, any other iterable is
    walked.  Together the ranks cover every batch exactly once, with the loader's own indices (the all-gather sorts by them)."""
    from simpleimagecaptionzoo_amd.engine import _rank_batches

    class Indexable:
        def __init__(self, n):
            self.n, self.touched = n, []

        def __len__(self):
            return self.n

        def __getitem__(self, i):
            self.touched.append(i)
            return ("batch", i)

    class Sharded:
        def shard(self, rank, world):
            return iter([(rank, "mine")])

    for world in (1, 2, 3, 8):
        seen = []
        for r in range(world):
            ld = Indexable(11)
            got = list(_rank_batches(ld, r, world))
            assert [i for i, _ in got] == list(range(r, 11, world)) and all(b == ("batch", i) for i, b in got)
            assert ld.touched == list(range(r, 11, world))            # nothing else was loaded
            seen += [i for i, _ in got]
        assert sorted(seen) == list(range(11))
    assert list(_rank_batches(Sharded(), 1, 2)) == [(1, "mine")]
    gen = (("g", i) for i in range(7))                                # a plain iterable: walked in full, foreign batches dropped
    assert list(_rank_batches(gen, 1, 3)) == [(1, ("g", 1)), (4, ("g", 4))]


def test_cooked_reference_blocks_equal_the_per_image_form():
    """ReferenceCooker.cook_images(as_block=True) -- what the device store appends since round 3 -- holds exactly the per-image
    arrays of the classic form, concatenated, with reference-level entry pointers."""
    from simpleimagecaptionzoo_amd.ciderd import ReferenceCooker
    from simpleimagecaptionzoo_amd.synth import document_frequency, synthetic_references
    from simpleimagecaptionzoo_amd.vocab import synthetic_vocab
    vocab = synthetic_vocab(211)
    words = [vocab.ix2word[i] for i in range(211)]
    dfd = document_frequency(synthetic_references(60, words, seed=3))
    ck = ReferenceCooker(dfd["document_frequency"], dfd["ref_len"], vocab.word2ix)
    refs = synthetic_references(9, words, seed=4)
    refs = [list(refs[i]) for i in range(9)]
    refs[2] = refs[2][:2] + ["zzz unknown words here"]               # out-of-vocabulary words, fewer references
    per = ck.cook_images(refs)
    blk = ck.cook_images(refs, as_block=True)
    assert blk["nref"].tolist() == [len(r) for r in refs]
    assert np.array_equal(blk["key"], np.concatenate([p[1] for p in per])) and np.array_equal(blk["ord"], np.concatenate([p[2] for p in per]))
    assert np.array_equal(blk["w"], np.concatenate([p[3] for p in per])) and np.array_equal(blk["norm"], np.concatenate([p[4] for p in per]))
    assert np.array_equal(blk["len"], np.concatenate([p[5] for p in per]))
    ends, e0 = [], 0
    for p in per:
        ends += (e0 + p[0][1:]).tolist()
        e0 += int(p[0][-1])
    assert blk["ep"][0] == 0 and blk["ep"][1:len(ends) + 1].tolist() == ends


def test_library_tokeniser_equals_str_split_and_falls_back_for_non_ascii():
    """icz_ciderd_cook_text tokenises like str.split() on ASCII text (runs of blanks, tabs, leading / trailing blanks, empty
    references) and shares the private ids of out-of-vocabulary words with the Python side; references with non-ASCII
    characters or an embedded newline take the Python tokeniser -- every array equal to the Python statement either way."""
    from simpleimagecaptionzoo_amd.ciderd import ReferenceCooker
    from simpleimagecaptionzoo_amd.synth import document_frequency, synthetic_references
    from simpleimagecaptionzoo_amd.vocab import synthetic_vocab
    v = synthetic_vocab(120)
    words = [v.ix2word[i] for i in range(120)]
    df = document_frequency(synthetic_references(80, words + ["oovA", "oovB"], seed=1))
    ck = ReferenceCooker(df["document_frequency"], df["ref_len"], v.word2ix)
    ascii_refs = [["  w5   w6\tw7 ", "", "w5 w5 w5", "newword w5 newword", "   "], ["w9"], ["oovA oovB w1 zzz"]]
    other_refs = [["w5 café w6", "w7 w8"], ["w1\nw2 w3"]]          # non-ASCII word, non-breaking space (a str.split() blank), newline
    for refs in (ascii_refs, other_refs, ascii_refs + other_refs):
        want = [ck.cook_image(r) for r in refs]
        got = ck.cook_images(refs)
        for a, b in zip(want, got):
            for x, y in zip(a, b):
                assert x.dtype == y.dtype and x.shape == y.shape and np.array_equal(x, y)
    # the same out-of-vocabulary word has one id on both sides
    assert ck._word_id("newword") == ck._word_id("newword") >= 120
    k = ck.cook_images([["newword"]])[0][1]
    assert k[0, 0] == ck._word_id("newword")


def test_bench_launcher_reports_dead_ranks_instead_of_hanging():
    """`python bench.py --gpus 2` with no launcher on a box without a GPU: both child ranks die at once (no device); the parent
    must not wait for a rendezvous -- it polls its children, collects their output, prints every rank's tail and exits non-zero
    within seconds, with no JSON line on stdout (bench.py: spawn_ranks)."""
    import subprocess
    import sys
    import time
    if __import__("torch").cuda.is_available():
        pytest.skip("needs a box without a GPU (the GPU variant is tests/test_gpu_bench_ranks.py::test_bench_launcher_kills_the_other_ranks_when_one_dies)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["ICZ_BENCH_RANK_TIMEOUT"] = "120"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--headline-only"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode != 0 and time.time() - t0 < 200
    assert "rank exit codes" in r.stderr and "---- rank 0" in r.stderr and "---- rank 1" in r.stderr and "No HIP GPUs" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_reads_its_committed_profiles_for_both_gemm_kernels():
    """bench.py's secondary roofline entries come from the committed rocprofv3 summaries under profiles/: since round 5 the many-row NT
    products sit on two kernels (the 128 x 128 two-barrier one and gemm_big_x3_kernel) -- the entries must find both and stay finite."""
    import bench
    assert bench._name_match("gemm_big_x3_kernel<*true, true>", "void icz::(anonymous namespace)::gemm_big_x3_kernel<128, 128, 2, 2, 3, false, true, true>(icz::GemmArgs, int, int)")
    assert not bench._name_match("gemm_big_x3_kernel<*true, true>", "void icz::(anonymous namespace)::gemm_big_x3_kernel<128, 128, 2, 2, 3, false, true, false>(icz::GemmArgs)")
    r = bench.csv_roofline("beam5_b128_kernel_stats.csv", bench.NT_BIG_KERNELS, "beam", flops_per_launch=2.0 * 640 * (4096 * 3072 + 4096 * 4096 + 10112 * 1024) / 3.0)
    assert "error" not in r and 0.2 < r["frac"] < 0.7, r
    a = bench.aoa_roofline(64)
    assert "error" not in a and 0.2 < a["frac"] < 0.7 and 10 <= a["launches_per_step"] <= 30, a
    us, src = bench.trace_avg_us("gemm_resident_x3_kernel")
    assert us and 10 < us < 30 and src.startswith("profiles/r0")


def test_routing_of_the_many_row_gemms_at_the_baseline_shapes():
    """gemm_big_cfg (host logic of csrc/gemm_big_x3.hip) at the shapes the BASELINE configs issue, as measured in EXPERIMENTS round 5
    section 6: the LSTM weight-gradient groups, predict's weight gradient, the XE vocabulary projection and the paired AoA Q/K/V
    projection on 256 x 256 tiles (1); the dgrad products over all steps and the AoA linear on 128 x 128 three per CU (4); beam-search
    steps and small outputs on the 128 x 128 two-barrier kernel (0)."""
    from simpleimagecaptionzoo_amd._lib import lib
    f = lib().icz_gemm_big_cfg_for
    NT, NN, TN = 0, 1, 2
    assert f(TN, 4096, 4096, 1280, 1) == 1 and f(TN, 4096, 3072, 1280, 1) == 1 and f(TN, 10112, 1024, 1280, 1) == 1
    assert f(TN, 4096, 2048, 1280, 1) == 0 and f(TN, 4096, 1024, 1280, 1) == 0
    assert f(NN, 1280, 1024, 10112, 8) == 4 and f(NN, 1280, 1024, 4096, 4) == 4 and f(NN, 640, 1024, 10112, 9) == 0
    assert f(NT, 640, 4096, 4096, 3) == 0 and f(NT, 320, 4096, 3072, 5) == 0 and f(NT, 640, 10112, 1024, 1) == 0
    assert f(NT, 1280, 10112, 1024, 1) == 1 and f(NT, 4608, 3072, 1024, 1) == 1
    assert f(NT, 2304, 2048, 2048, 1) == 4 and f(NT, 4608, 2048, 2048, 1) == 4 and f(NT, 2304, 1024, 1024, 1) == 0
    assert f(5, 1, 1, 1, 1) == -1


def test_bench_parses_what_rccl_says_about_its_rings(tmp_path):
    """bench.py --gpus N quotes rank 0's RCCL choices (NCCL_DEBUG=INFO into NCCL_DEBUG_FILE): channel count, algorithm / protocol per
    all-reduce size, environment overrides.  The wording below is NCCL 2.2x's; unknown lines are ignored, a missing file is an error
    entry -- never an exception (the headline must not depend on it)."""
    import bench
    log = tmp_path / "rccl.log"
    log.write_text("\n".join([
        "box:77:101 [0] NCCL INFO NCCL_DEBUG_SUBSYS set by environment to INIT,GRAPH,TUNING,ENV",
        "box:77:101 [0] NCCL INFO RCCL version 2.22.3+hip7.0",
        "box:77:101 [0] NCCL INFO Channel 00/16 :    0   1   2   3   4   5   6   7",
        "box:77:101 [0] NCCL INFO Channel 15/16 :    0   7   6   5   4   3   2   1",
        "box:77:101 [0] NCCL INFO Trees [0] 1/-1/-1->0->-1 [1] 1/-1/-1->0->-1",
        "box:77:101 [0] NCCL INFO Connected all rings",
        "box:77:101 [0] NCCL INFO 16 coll channels, 0 collnet channels, 0 nvls channels, 16 p2p channels, 2 p2p channels per peer",
        "box:77:101 [0] NCCL INFO comm 0x55 rank 0 nranks 8 cudaDev 0 busId 5000 commId 0xabc - Init COMPLETE",
        "box:77:101 [0] NCCL INFO AllReduce: 166395904 Bytes -> Algo 1 proto 2 time 1503.2",
        "box:77:101 [0] NCCL INFO AllReduce: 4 Bytes -> Algo 0 proto 0 time 12.1",
        "garbage that matches nothing",
    ]))
    r = bench.rccl_info(str(log))
    assert r["channels"] == 16 and r["coll_channels"] == 16
    assert r["algo_proto"] == {"AllReduce 166395904 B": "Ring / Simple", "AllReduce 4 B": "Tree / LL"}
    assert len(r["env"]) == 1 and any("Init COMPLETE" in l for l in r["lines"]) and r["n_lines"] == 11
    miss = bench.rccl_info(str(tmp_path / "nope.log"))
    assert "error" in miss and miss["channels"] is None
