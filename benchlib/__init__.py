"""bench.py's parts (the CLI stays `python bench.py ...`; the driver's contract is unchanged):

    control.py       the N > 1 control plane — a key/value store (torch.distributed.TCPStore) that survives a failed rank:
                     barriers, gathers and reductions count the ranks that are still alive; the self-launcher kills siblings
    common.py        the run context (rank, device, host placement, circuit), lanes, shared leg helpers
    segment.py       --config segment  (BASELINE config 2; the default line and its secondary legs)
    block.py         --config block    (configs 3 / 4; also the chained block)
    succinct.py      --config succinct (config 5)
    dev.py           --config dev      (config 1: RISC0_DEV_MODE plumbing, no GPU)
    roofline.py      roofline{} for the dominant op: live HIP-event durations, HBM traffic and VALU issue from rocprofv3 --pmc
    cpu_baseline.py  the CPU oracle timed on the host cores (the only place outside tests/ that touches oracle/)
"""
