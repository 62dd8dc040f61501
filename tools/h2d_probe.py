"""Host-side cost of the cross-stream hand-off of an H2D copy (DIAGNOSTIC): copy on the compute stream vs copy on a copy
stream + event record + wait_event, with a ~250 us kernel per step on 7 compute streams."""
import time, torch
dev = torch.device("cuda", 0)
S, K, N = 7, 300, 153557
streams = [torch.cuda.Stream() for _ in range(S)]
cs = torch.cuda.Stream()
host = [torch.randn(N, 6).pin_memory() for _ in range(4)]
stage = [[torch.empty(N, 6, device=dev) for _ in range(2)] for _ in range(S)]
flag_host = torch.arange(64, dtype=torch.int32).pin_memory()
flag_dev = torch.zeros(S, dtype=torch.int32, device=dev)
work = [torch.randn(2048, 2048, device=dev) for _ in range(S)]
def kern(k):
    for _ in range(3): work[k] @ work[k]
def run(mode):
    torch.cuda.synchronize()
    evc = [[torch.cuda.Event(), torch.cuda.Event()] for _ in range(S)]
    evf = [[None, None] for _ in range(S)]
    t0 = time.perf_counter(); tw = 0.0
    for i in range(K):
        k, slot = i % S, (i // S) & 1
        st = streams[k]
        a = time.perf_counter()
        if mode == "same":
            with torch.cuda.stream(st):
                stage[k][slot].copy_(host[i % 4], non_blocking=True)
        elif mode == "copystream":
            if evf[k][slot] is not None: cs.wait_event(evf[k][slot])
            with torch.cuda.stream(cs):
                stage[k][slot].copy_(host[i % 4], non_blocking=True)
                evc[k][slot].record(cs)
            st.wait_event(evc[k][slot])
        elif mode == "cs_copy_only":
            with torch.cuda.stream(cs):
                stage[k][slot].copy_(host[i % 4], non_blocking=True)
        elif mode == "cs_copy_record":
            with torch.cuda.stream(cs):
                stage[k][slot].copy_(host[i % 4], non_blocking=True)
                evc[k][slot].record(cs)
        elif mode == "cs_copy_flagcopy":
            with torch.cuda.stream(cs):
                stage[k][slot].copy_(host[i % 4], non_blocking=True)
                flag_dev[k:k+1].copy_(flag_host[i % 64: i % 64 + 1], non_blocking=True)
        elif mode == "rec_wait_only":
            evc[k][slot].record(cs)
            st.wait_event(evc[k][slot])
        elif mode == "copystream_hostsync":
            if evf[k][slot] is not None: evf[k][slot].synchronize()
            with torch.cuda.stream(cs):
                stage[k][slot].copy_(host[i % 4], non_blocking=True)
                evc[k][slot].record(cs)
            st.wait_event(evc[k][slot])
        tw += time.perf_counter() - a
        with torch.cuda.stream(st):
            kern(k)
            if mode in ("copystream", "copystream_hostsync"):
                if evf[k][slot] is None: evf[k][slot] = torch.cuda.Event()
                evf[k][slot].record(st)
    issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print(f"{mode:22s} issue {issue/K*1e3:.3f} ms/step (copy part {tw/K*1e3:.3f})  total {tot/K*1e3:.3f} ms/step")
for m in ("none", "same", "cs_copy_only", "cs_copy_record", "cs_copy_flagcopy", "rec_wait_only", "copystream", "same"):
    run(m)
