"""Development aid: patches lx_ckpt.hip IN PLACE with cycle counters around the phases of ckpt_backtrace_kernel (printed
every 8th launch).  Never commit the patched file: `git checkout lambda_amd/csrc/lx_ckpt.hip` afterwards."""
import sys
p = sys.argv[1] if len(sys.argv) > 1 else 'lambda_amd/csrc/lx_ckpt.hip'
s = open(p).read()


def rep(a, b, must=True):
    global s
    if s.count(a) != 1:
        if must:
            raise SystemExit("anchor not found / not unique: " + a[:70])
        return
    s = s.replace(a, b, 1)


rep('''template <int G, int C>
__global__ __launch_bounds__(64, LX_BT_WAVES) void ckpt_backtrace_kernel(TraceParams p)
{''', '''__device__ unsigned long long bt_prof[16];
#define PROF_T() ((unsigned long long)__builtin_readcyclecounter())
template <int G, int C>
__global__ __launch_bounds__(64, LX_BT_WAVES) void ckpt_backtrace_kernel(TraceParams p)
{
    unsigned long long pf_short = 0, pf_refill = 0, pf_dp = 0, pf_walk = 0, pf_nshort = 0, pf_can = 0, pf_ntile = 0, pf_tneed = 0, pf_nrefill = 0, pf_t0 = PROF_T(), pf_setup = 0, pf_steps = 0;''')
rep('''        if (n_idle > 0 && (run_tile || n_idle >= refill_at))
        {''', '''        if (n_idle > 0 && (run_tile || n_idle >= refill_at))
        {
            unsigned long long const pt = PROF_T(); ++pf_nrefill;''')
rep('''            run_tile = n_can == 0 || n_tile >= tile_at;
        }

        if (!run_tile)
        {''', '''            run_tile = n_can == 0 || n_tile >= tile_at;
            pf_refill += PROF_T() - pt;
        }

        if (!run_tile)
        {
            unsigned long long const pt = PROF_T(); ++pf_nshort; pf_can += n_can;''')
rep('''                        if (!take_diagonal(hk[h], dec(hw[h] & 0xffffu), true, hqw[h], hsw[h]))
                            blocked = true;
            }
            continue;
        }
''', '''                        if (!take_diagonal(hk[h], dec(hw[h] & 0xffffu), true, hqw[h], hsw[h]))
                            blocked = true;
            }
            pf_short += PROF_T() - pt;
            continue;
        }
        unsigned long long const ptile0 = PROF_T(); ++pf_ntile; pf_tneed += n_tile;
''')
rep('''        bool     rows_left = true;
''', '''        unsigned long long const ptile1 = PROF_T(); pf_setup += ptile1 - ptile0;
        bool     rows_left = true;
''')
rep('''        bool walk_ok = true;
''', '''        unsigned long long const ptile2 = PROF_T(); pf_dp += ptile2 - ptile1;
        bool walk_ok = true;
''')
rep('''            emit(diag ? (uint32_t)'M' : (vert ? (uint32_t)'D' : (uint32_t)'I'));''', '''            emit(diag ? (uint32_t)'M' : (vert ? (uint32_t)'D' : (uint32_t)'I'));
            ++pf_steps;''')
rep('''        blocked = false;
        } // tile_need
    }
}''', '''        blocked = false;
        pf_walk += PROF_T() - ptile2;
        } // tile_need
    }
    {
        auto wmax = [](unsigned long long x) { for (int o = 32; o > 0; o >>= 1) { unsigned long long y = ((unsigned long long)(uint32_t)__shfl_xor((int)(x >> 32), o) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)x, o); x = x > y ? x : y; } return x; };
        pf_setup = wmax(pf_setup); pf_dp = wmax(pf_dp); pf_walk = wmax(pf_walk);
    }
    atomicAdd(&bt_prof[12], pf_steps);
    if (lane == 0)
    {
        unsigned long long const v[12] = {PROF_T() - pf_t0, pf_short, pf_refill, pf_setup, pf_dp, pf_walk, pf_nshort, pf_can, pf_ntile, pf_tneed, pf_nrefill, 1};
        for (int x = 0; x < 12; ++x)
            atomicAdd(&bt_prof[x], v[x]);
    }
}''')
rep('''    hipError_t e = hipMemsetAsync(p.work_counter, 0, sizeof(uint32_t), stream);''', '''    {
        static int calls = 0;
        if (++calls % 8 == 0)
        {
            (void)hipDeviceSynchronize();
            unsigned long long v[16];
            (void)hipMemcpyFromSymbol(v, HIP_SYMBOL(bt_prof), sizeof(v));
            double const w = (double)v[11];
            fprintf(stderr, "[bt_prof] waves %.0f per-wave cycles: total %.0f short %.0f refill %.0f setup %.0f dp %.0f walk %.0f | passes: short %.1f (lanes %.1f) tile %.1f (lanes %.1f) refill %.1f | walk steps per wave (all lanes) %.0f\\n",
                    w, v[0] / w, v[1] / w, v[2] / w, v[3] / w, v[4] / w, v[5] / w, v[6] / w, v[6] ? (double)v[7] / v[6] : 0, v[8] / w, v[8] ? (double)v[9] / v[8] : 0, v[10] / w, v[12] / w);
            unsigned long long z[16] = {};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(bt_prof), z, sizeof(z));
        }
    }
    hipError_t e = hipMemsetAsync(p.work_counter, 0, sizeof(uint32_t), stream);''')
if '#include <cstdio>' not in s:
    s = s.replace('#include <cstdlib>', '#include <cstdio>\n#include <cstdlib>')
open(p, 'w').write(s)
