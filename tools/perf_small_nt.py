"""dev tool (round 5): SCST step at 8 / 16 / 32 images with the BPTT dgrad products of <= 32 rows on the NN kernel (option small_nt = 0)
and on the transposed weight copies through the fp32 NT kernel (1, default); alternating legs in one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

for B in (8, 16, 32):
    eng, opt, vocab, words = bench.build_engine("cuda:0", 2 * B)
    batches = bench.make_batches(6, B, words, "cuda:0", 0)
    for bt in batches:
        eng.scorer().preload(bt[2])
    h = eng._hot_handle()
    for rnd in range(3):
        for on in (0, 1):
            h.set_option("small_nt", on)
            eng.SCST_training_epoch(batches, opt, None, tqdm_visible=False)
            eng.phase_events = []
            eng.SCST_training_epoch(batches * 3, opt, None, tqdm_visible=False)
            torch.cuda.synchronize()
            ph = eng.phase_times(skip=2)
            print("b=%-3d small_nt=%d rollouts %.3f backward %.3f adam %.3f  sum %.3f" % (B, on, ph["rollouts"], ph["backward"], ph["adam"],
                                                                                   ph["rollouts"] + ph["reward"] + ph["backward"] + ph["adam"]), flush=True)
    del eng, opt
