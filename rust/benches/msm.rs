// `cargo bench` in one command (plain `harness = false` binary: no criterion, so nothing beyond the crate's own dependencies):
// the ZPrize workload -- init untimed, then 4 batches of 2^BENCH_NPOW scalars from host memory through multi_scalar_mult -- and,
// beside it, REAL ark-ec `VariableBaseMSM::multi_scalar_mul` on this host's cores for one batch, compared with the GPU's result.
// bench.py's `cpu_baseline.ark_ec` probe runs this binary when a cargo toolchain is present and parses the two KEY=VALUE lines.
// (The reference's own criterion bench, P1A combined-top-solutions/benches/msm.rs, also runs unchanged against this crate:
//  INTEGRATION.md section 1 says which one line of it to edit.)
use ark_bls12_377::G1Affine;
use ark_ec::msm::VariableBaseMSM;
use ark_ec::ProjectiveCurve;
use ark_ff::BigInteger256;
use std::str::FromStr;
use std::time::Instant;

use mi355_msm::*;

fn env_usize(name: &str, default: usize) -> usize {
    std::env::var(name).ok().and_then(|v| usize::from_str(&v).ok()).unwrap_or(default)
}

fn main() {
    let npow = env_usize("BENCH_NPOW", 26);
    let reps = env_usize("BENCH_REPS", 3);
    let batches = 4;
    let n = 1usize << npow;
    let (points, scalars) = util::generate_points_scalars::<G1Affine>(n, batches);
    let bigints = unsafe { std::mem::transmute::<&[_], &[BigInteger256]>(scalars.as_slice()) };

    let mut context = multi_scalar_mult_init(points.as_slice()); // untimed, as in the reference bench
    let mut results = multi_scalar_mult(&mut context, points.as_slice(), bigints); // warm-up
    let t = Instant::now();
    for _ in 0..reps {
        results = multi_scalar_mult(&mut context, points.as_slice(), bigints);
    }
    let gpu_ms = t.elapsed().as_secs_f64() * 1e3 / reps as f64;
    println!("MI355_MSM_MS_PER_{}_BATCHES={:.3} NPOW={}", batches, gpu_ms, npow);

    if env_usize("BENCH_ARK_EC", 1) != 0 {
        let threads = std::thread::available_parallelism().map(|v| v.get()).unwrap_or(1);
        let t = Instant::now();
        let expected = VariableBaseMSM::multi_scalar_mul(points.as_slice(), &bigints[..n]);
        let cpu_ms = t.elapsed().as_secs_f64() * 1e3;
        println!("ARK_EC_CPU_MS={:.1} NPOW={} HOST_THREADS={}", cpu_ms, npow, threads);
        assert_eq!(results[0].into_affine(), expected.into_affine(), "GPU result differs from ark-ec");
        println!("GPU_EQUALS_ARK_EC=1");
    }
}
