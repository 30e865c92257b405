//! Input generator of the harness shape: a small set of random subgroup points replicated by doubling the vector (which
//! also exercises the doubling case of the group law), a sprinkled point at infinity, uniform scalars.
//! Behaviour of P1A 6block/src/util.rs:10-37; written against the same arkworks 0.3 API.
use ark_ec::{AffineCurve, ProjectiveCurve};
use ark_std::UniformRand;
use rand::SeedableRng;
use rand_chacha::ChaCha20Rng;

pub fn generate_points_scalars<G: AffineCurve>(len: usize, batch_size: usize) -> (Vec<G>, Vec<G::ScalarField>) {
    const DISTINCT: usize = 1 << 11;
    let mut rng = ChaCha20Rng::from_entropy();
    let projective: Vec<G::Projective> = (0..DISTINCT.min(len.max(1))).map(|_| G::Projective::rand(&mut rng)).collect();
    let mut points = <G::Projective as ProjectiveCurve>::batch_normalization_into_affine(&projective);
    if points.len() > 3 {
        points[3] = G::zero();
    }
    while points.len() < len {
        let mut again = points.clone();
        points.append(&mut again);
    }
    points.truncate(len);
    let scalars = (0..len * batch_size).map(|_| G::ScalarField::rand(&mut rng)).collect();
    (points, scalars)
}
