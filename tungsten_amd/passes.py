"""The render loop around one Context: StandaloneRenderer's `while (!done) startRender/waitForCompletion`
(src/tungsten/Shared.hpp:283-300) with PathTraceIntegrator's work generation (PathTraceIntegrator.cpp:110-134,
220-239).  The C++ mirror of the same loop is tungsten_amd/host/integrator.cpp; this one serves the Python
harness (bench.py, multi-process launches)."""
from ._host import AdaptiveScheduler


def render_passes(ctx, res_x, res_y, spp, spp_step=16, adaptive=False, seed=0xBA5EBA11, exchange=None):
    """Render `spp` samples per pixel (on average, when adaptive) on `ctx` in passes of `spp_step`.

    exchange: with more than one rank and adaptive sampling, a callable merging the ranks' SampleRecord
    tables (tungsten_amd.dist.exchange_records); uniform passes need no exchange at all.
    Returns the AdaptiveScheduler (None for uniform sampling).
    """
    if not adaptive:
        ctx.render(spp, spp_step)
        return None
    work = AdaptiveScheduler(res_x, res_y, seed)
    current = 0
    while current < spp:
        nxt = min(current + spp_step, spp)
        if work.needs_statistics(True, current):
            ctx.wait()
            table = ctx.download_records()
            if exchange is not None:
                table = exchange(table)
            work.set_statistics(table)
        job = work.generate(current, nxt, True)
        if job is not None:  # None: converged, the pass renders nothing (PathTraceIntegrator.cpp:229-232)
            ctx.render_pass_adaptive(*job)
        current = nxt
    ctx.wait()
    return work
