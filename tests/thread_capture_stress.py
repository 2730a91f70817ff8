"""Two fit threads of one process capturing their epoch graphs while the sibling sets up, reads back, predicts and tears down
(tests/test_gpu_parity.py runs this in a subprocess; round 5, train.DEVICE_LOCK).  Prints "OK 50 fits" or raises."""
import gc
import os
import sys
import threading
import traceback

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locator_amd.net import LocatorNet, upload_genotypes  # noqa: E402
from locator_amd.train import DEVICE_LOCK, fit  # noqa: E402


def main(n_fits=25):
    rng = np.random.default_rng(0)
    x = rng.integers(0, 3, (160, 640)).astype(np.uint8)
    y = rng.normal(size=(160, 2)).astype(np.float32)
    X, Y = upload_genotypes(x), torch.from_numpy(y).cuda()
    rows = torch.arange(128, 160, dtype=torch.int32, device="cuda")

    def one_fit(rep):
        with DEVICE_LOCK:                                   # what locator._fit_unit does around a unit
            net = LocatorNet(X, Y, 640, 256, 10, 0.25, seed=7, replicate=rep)
            h = fit(net, np.arange(96), np.arange(96, 128), max_epochs=6, patience=6)     # releases the lock while it loops
            yh = torch.zeros((32, 2), device="cuda")
            net.predict_rows(rows, 32, yh)
            torch.cuda.current_stream().synchronize()
            out = (h.history["val_loss"], yh.cpu().numpy())
            del net, h
            gc.collect()                                    # graphs / events of the finished fit die here, under the lock
            return out

    ref = {rep: one_fit(rep) for rep in range(4)}
    got, errs = {}, []

    def worker(tid):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for i in range(n_fits):
                    rep = (tid * n_fits + i) % 4
                    got[(tid, i)] = (rep, one_fit(rep))
        except Exception:                                   # noqa: BLE001
            errs.append(traceback.format_exc())

    th = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise SystemExit(errs[0])
    assert len(got) == 2 * n_fits
    for (tid, i), (rep, (vl, yh)) in got.items():
        assert vl == ref[rep][0] and np.array_equal(yh, ref[rep][1]), (tid, i, rep)
    print(f"OK {2 * n_fits} fits")


if __name__ == "__main__":
    main()
