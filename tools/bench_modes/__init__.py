"""bench.py's modes that the driver's default command reaches only through a leg or a flag:
`sharded` (--mode sharded: configs[2] - [4] shaped runs), `cpu` (the cpu_baseline leg and its parity replay).
bench.py imports them; they reach bench.py's configuration (FOV, DEPTH, ...: set by configure()) as B.NAME."""
