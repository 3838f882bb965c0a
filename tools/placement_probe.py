#!/usr/bin/env python3
"""Does streaming bandwidth depend on where in HBM the arrays live?  Allocates 8 chunks of 30 GiB and times
c = a + b (2 reads, 1 write, 4 GiB each) with the three arrays (i) inside one chunk, for every chunk, and
(ii) in three different chunks."""
import time

import torch


def main():
    G = 1 << 27  # doubles per GiB
    chunks = [torch.zeros(30 * G, dtype=torch.float64, device="cuda") for _ in range(8)]
    print("chunk base addresses (GiB):", [round(c.data_ptr() / 2 ** 30, 1) for c in chunks])

    def bw(a, b, c):
        for _ in range(2):
            torch.add(a, b, out=c)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            torch.add(a, b, out=c)
        torch.cuda.synchronize()
        return 3 * a.numel() * 8 * 5 / (time.perf_counter() - t0) / 1e12

    n = 4 * G
    for k, ch in enumerate(chunks):
        print(f"all three in chunk {k}: {bw(ch[0:n], ch[8 * G : 8 * G + n], ch[16 * G : 16 * G + n]):.2f} TB/s", flush=True)
    for (i, j, k) in ((0, 1, 2), (0, 3, 6), (2, 4, 7), (5, 6, 7), (1, 1, 5)):
        print(f"a in {i}, b in {j}, c in {k}: {bw(chunks[i][0:n], chunks[j][8 * G : 8 * G + n], chunks[k][16 * G : 16 * G + n]):.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
