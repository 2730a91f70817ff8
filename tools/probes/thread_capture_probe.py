"""Two fits on two threads / streams of one process, each capturing its epoch graph while the other keeps working: which
HIP stream-capture mode tolerates that?  (round 4; replicates.py fit threads)   python3 tools/probes/thread_capture_probe.py MODE"""
import os
import sys
import threading
import time
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "thread_local"
    import torch

    import locator_amd.train as T
    from locator_amd.net import LocatorNet, upload_genotypes
    T.CAPTURE_ERROR_MODE = mode
    rng = np.random.default_rng(0)
    x = rng.integers(0, 3, (500, 20000)).astype(np.uint8)
    y = rng.normal(size=(500, 2)).astype(np.float32)
    X, Y = upload_genotypes(x), torch.from_numpy(y).cuda()
    out, errs = {}, []

    def work(i):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                net = LocatorNet(X, Y, 20000, 256, 10, 0.25, seed=1, replicate=i)
                h = T.fit(net, np.arange(400), np.arange(400, 450), max_epochs=40, patience=100)
                yh = torch.zeros((50, 2), device="cuda")
                net.predict_rows(torch.arange(450, 500, dtype=torch.int32, device="cuda"), 50, yh)
                if len(sys.argv) > 2 and sys.argv[2] == "devsync":
                    torch.cuda.synchronize()                 # a DEVICE-wide wait while the other thread may be capturing
                _ = torch.empty(1 << 20, dtype=torch.uint8).pin_memory()   # what the window loader thread does meanwhile
                s.synchronize()
                out[i] = (h.history["val_loss"], yh.cpu().numpy())
        except Exception:                                   # noqa: BLE001
            errs.append(traceback.format_exc())

    for rep in range(3):
        t0 = time.time()
        th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        print(f"mode {mode} round {rep}: {time.time() - t0:.2f} s, errors {len(errs)}", flush=True)
        if errs:
            print(errs[0][-1500:])
            break
    if not errs:
        ref = dict(out)
        out.clear()
        work(0); work(1)                                    # the same two fits one after the other: same results?
        same = all(ref[i][0] == out[i][0] and np.array_equal(ref[i][1], out[i][1]) for i in range(2))
        print(f"mode {mode}: concurrent == sequential results: {same}")


if __name__ == "__main__":
    main()
