// The harness's correctness test against arkworks on the CPU, in the shape of
// P1A combined-top-solutions/tests/msm.rs:15-40: 4 batches over one base vector, every batch compared in affine form.
// TEST_NPOW selects the size (default 2^16 here; the harness default is 2^26 and takes minutes of CPU time).
use ark_bls12_377::G1Affine;
use ark_ec::msm::VariableBaseMSM;
use ark_ec::ProjectiveCurve;
use ark_ff::BigInteger256;
use std::str::FromStr;

use mi355_msm::*;

#[test]
fn msm_correctness() {
    let npow = i32::from_str(&std::env::var("TEST_NPOW").unwrap_or("16".to_string())).unwrap();
    let batches = 4;
    let (points, scalars) = util::generate_points_scalars::<G1Affine>(1usize << npow, batches);

    let mut context = multi_scalar_mult_init(points.as_slice());
    // `Fr` values are handed over as their internal representation, as the harness does
    let bigints = unsafe { std::mem::transmute::<&[_], &[BigInteger256]>(scalars.as_slice()) };
    let results = multi_scalar_mult(&mut context, points.as_slice(), bigints);
    assert_eq!(results.len(), batches);

    for b in 0..batches {
        let slice = &bigints[b * points.len()..(b + 1) * points.len()];
        let expected = VariableBaseMSM::multi_scalar_mul(points.as_slice(), slice).into_affine();
        assert_eq!(results[b].into_affine(), expected, "batch {}", b);
    }
}

#[test]
fn trait_shaped_entry_points() {
    use ark_bls12_377::Fr;
    use ark_ff::PrimeField;
    let (points, scalars) = util::generate_points_scalars::<G1Affine>(1usize << 10, 1);
    let bigints: Vec<_> = scalars.iter().map(|s: &Fr| s.into_repr()).collect();
    let expected = VariableBaseMSM::multi_scalar_mul(points.as_slice(), bigints.as_slice());
    assert_eq!(variable_base::msm(&points, &scalars).into_affine(), expected.into_affine());
    assert_eq!(variable_base::msm_bigint(&points, &bigints).into_affine(), expected.into_affine());
    // length mismatch: msm chops, msm_checked reports the shorter length
    assert_eq!(variable_base::msm_checked(&points[..1000], &scalars), Err(1000));
    let chopped = VariableBaseMSM::multi_scalar_mul(&points[..1000], &bigints[..1000]);
    assert_eq!(variable_base::msm(&points[..1000], &scalars).into_affine(), chopped.into_affine());
}
