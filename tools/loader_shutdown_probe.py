"""How long does a DataLoader with worker processes take to shut down next to an initialised HIP runtime? (the Trainer tests spent 20 s per Trainer there)"""
import sys
import time

import torch


class DS(torch.utils.data.Dataset):
    def __len__(self):
        return 64

    def __getitem__(self, i):
        return torch.full((80, 100), float(i))


if __name__ == "__main__":
    torch.zeros(1, device="cuda")
    for ctx in (None, "forkserver", "spawn"):
        for pin in (False, True):
            for pers in (True, False):
                t0 = time.time()
                dl = torch.utils.data.DataLoader(DS(), batch_size=4, num_workers=4, pin_memory=pin, persistent_workers=pers, prefetch_factor=4,
                                                 multiprocessing_context=ctx)
                it = iter(dl)
                for _ in range(3):
                    next(it)
                t1 = time.time()
                del it, dl
                import gc
                gc.collect()
                print(f"context={ctx} pin={pin} persistent={pers}: start + 3 batches {t1 - t0:.2f} s, shutdown {time.time() - t1:.2f} s", flush=True)
