"""bench.py's implementation, split by concern: common (logging, the one JSON line, hashes), launch (self-spawn,
CPU dry run), transports (halo bring-up, first contact, PEER / RCCL comparison, proxy, scatter timing), roofline (bytes contracts,
PMC replay, roofline.secondary), cpu_legs (the oracle as CPU baseline and as checker), steps (RK4 / Westervelt step lines),
aux_lines (the N = 1 auxiliary lines), harvest (the N > 1 secondary lines), apply (the headline and the other apply modes)."""
