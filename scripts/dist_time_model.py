"""Paper time model of the 2-D block-cyclic factorisation (gpexp_amd/dist.py) from single-GPU measurements.
NOT a measurement: RCCL with more than one rank has never run (one GPU per lease).

Inputs measured on one MI355X (scripts/probe_small_potrf.py, profiles/r02_dist_model_inputs.txt):
  potrf(nb) of a diagonal block; rate of a rank-nb lower update; panel-solve rate ~40 TF/s (short-K products);
assumed: xGMI 153 GB/s per link and direction, 7 links per GPU; 25 us per collective hand-off / dependent launch group.
Per step k (h = rows below the diagonal block), the three concurrent strands of dist2_potrf:
  diag   = potrf(nb) + solve of the ONE block L[k+1,k] + its broadcast along a process row + update of diagonal block k+1
           (the critical-path-first chain: what diag(k+1) really waits for)
  panel  = diagonal-block broadcast down the process column + panel solve of h/Pr rows + all-link panel broadcast
           (2 phases, each piece/(W-1) bytes per link; 1 phase for 2 ranks) + look-ahead column update
  update = h^2 nb / W flops at the rank-nb rate (every rank's share of the trailing update)
The factorisation takes about max( sum_k diag_k , sum_k max(panel_k, update_k) ); "serial" = what it would take with the
diagonal waiting for the whole panel (sum_k max(diag_k + panel_k, update_k)), the schedule before the chain was split."""
import sys
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
T_POTRF = {256: 0.118e-3, 512: 0.236e-3, 1024: 0.465e-3}
RATE = {256: 49.9e12, 512: 57.3e12, 1024: 64.0e12}
TRSM_RATE, LINK, LAT = 40e12, 153e9, 25e-6
T1 = 0.190   # single-GPU potrf at N = 32768 (profiles/r02_potrf_variants.txt); scaled by N^3 for other sizes
print("N = %d; single GPU %.1f ms" % (N, 1e3 * T1 * (N / 32768.0) ** 3))
print("%-6s %-5s %5s | %8s %8s %9s %8s | %9s %8s | %9s" % ("ranks", "grid", "nb", "diag ms", "panel ms", "update ms", "comm ms",
                                                         "total ms", "speed-up", "serial ms"))
for W, (Pr, Pc) in ((2, (1, 2)), (4, (2, 2)), (8, (2, 4))):
    for nb in (256, 512, 1024):
        nblk = N // nb
        diag_sum = panel_sum = upd_sum = comm_sum = pu = serial = 0.0
        for k in range(nblk):
            h = N - (k + 1) * nb
            piece = (h / Pr) * nb * 8.0
            phases = 1.0 if W == 2 else 2.0
            comm = phases * piece / ((W - 1) * LINK) + LAT if W > 1 else 0.0
            dbc = (nb * nb + nb * 128) * 8.0 / LINK + LAT if Pr > 1 else 0.0
            early = (nb * nb * 8.0 / LINK + LAT) if Pc > 1 else 0.0
            diag = T_POTRF[nb] + LAT + nb ** 3 / TRSM_RATE + early + 2.0 * nb ** 3 / RATE[nb] + LAT
            panel = dbc + (h / Pr) * nb * nb / TRSM_RATE + LAT + comm + 2.0 * (h / Pr) * nb * nb / RATE[nb] + LAT
            upd = h * h * nb / W / RATE[nb]
            diag_sum += diag
            panel_sum += panel
            upd_sum += upd
            comm_sum += comm
            pu += max(panel, upd)
            serial += max(diag + panel, upd)
        tot = max(diag_sum, pu)
        t1 = T1 * (N / 32768.0) ** 3
        print("%-6d %dx%-3d %5d | %8.1f %8.1f %9.1f %8.1f | %9.1f %7.1fx | %9.1f" % (W, Pr, Pc, nb, 1e3 * diag_sum, 1e3 * panel_sum,
              1e3 * upd_sum, 1e3 * comm_sum, 1e3 * tot, t1 / tot, 1e3 * serial))
