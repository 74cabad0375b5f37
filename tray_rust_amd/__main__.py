"""`python -m tray_rust_amd --worker [-n <number>]`: the reference's `tray_rust --worker` (src/main.rs:24,148-166) on an MI355X --
waits on port 63234 for a tray_rust master, renders the slice of the tile queue it is told to, sends every frame back.
-n (threads) is accepted and ignored: the GPU renders. --device picks the GPU, --seed the render's seed (the reference seeds from
the OS; here samples are keyed by pixel and sample index, so workers of one frame may share a seed)."""
import argparse
import sys

from . import Hip, distrib


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m tray_rust_amd", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--worker", action="store_true", help="start a worker process that listens for a master (the only mode: single-node "
                    "rendering and the master stay with the reference's CLI / bench.py)")
    ap.add_argument("-n", type=int, default=None, metavar="<number>", help="threads (ignored)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--port", type=int, default=distrib.PORT, help="(tests only; the reference's master always connects to 63234)")
    args = ap.parse_args(argv)
    if not args.worker:
        ap.error("only --worker is built (SURVEY 8f rank 2); render single frames through tray_rust_amd.Hip or bench.py")
    distrib.worker_node(Hip(args.device, seed=args.seed), args.n or 1, port=args.port)
    return 0


if __name__ == "__main__":
    sys.exit(main())
