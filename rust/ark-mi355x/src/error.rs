//! Error type of the backend: `SynthesisError` where the reference has a matching variant
//! (relations/src/utils/error.rs:5-21), and DISTINCT variants for device-side failures -- a HIP fault, an RCCL
//! failure or an out-of-memory condition must never read as "the circuit is unsatisfiable".
use ark_relations::gr1cs::SynthesisError;
use core::fmt;

use crate::ffi;

#[derive(Clone, Debug, PartialEq, Eq)]
pub enum Mi355xError {
    /// A synthesis-level error, either raised by the circuit itself or reported by the library
    /// (`AssignmentMissing`, `Unsatisfiable`, `PolynomialDegreeTooLarge`).
    Synthesis(SynthesisError),
    /// ARK355_EINVAL: malformed arguments (wrong lengths, mismatched key / matrices, bad serialized point ...).
    InvalidArgument(String),
    /// ARK355_ENOMEM: host or HBM allocation failed.
    OutOfMemory(String),
    /// ARK355_EHIP: a HIP runtime call or kernel launch failed.
    Hip(String),
    /// ARK355_ERCCL: communicator creation or a collective failed.
    Rccl(String),
    /// ARK355_ENODEV: no MI355X visible to the process.
    NoDevice,
    /// A code this version of the shim does not know.
    Unknown(i32, String),
}

impl Mi355xError {
    /// Map a non-zero return code (plus `ark355_last_error`) to an error value.
    pub fn from_code(rc: i32, msg: String) -> Self {
        match rc {
            ffi::ARK355_E_ASSIGNMENT_MISSING => Self::Synthesis(SynthesisError::AssignmentMissing),
            ffi::ARK355_E_UNSATISFIABLE => Self::Synthesis(SynthesisError::Unsatisfiable),
            ffi::ARK355_E_POLY_DEGREE_TOO_LARGE => Self::Synthesis(SynthesisError::PolynomialDegreeTooLarge),
            ffi::ARK355_EINVAL => Self::InvalidArgument(msg),
            ffi::ARK355_ENOMEM => Self::OutOfMemory(msg),
            ffi::ARK355_EHIP => Self::Hip(msg),
            ffi::ARK355_ERCCL => Self::Rccl(msg),
            ffi::ARK355_ENODEV => Self::NoDevice,
            other => Self::Unknown(other, msg),
        }
    }
}

impl From<SynthesisError> for Mi355xError {
    fn from(e: SynthesisError) -> Self {
        Self::Synthesis(e)
    }
}

impl fmt::Display for Mi355xError {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        match self {
            Self::Synthesis(e) => write!(f, "{e}"),
            Self::InvalidArgument(m) => write!(f, "libark355: invalid argument: {m}"),
            Self::OutOfMemory(m) => write!(f, "libark355: out of memory: {m}"),
            Self::Hip(m) => write!(f, "libark355: HIP error: {m}"),
            Self::Rccl(m) => write!(f, "libark355: RCCL error: {m}"),
            Self::NoDevice => write!(f, "libark355: no MI355X device"),
            Self::Unknown(rc, m) => write!(f, "libark355: error {rc}: {m}"),
        }
    }
}

impl ark_std::error::Error for Mi355xError {}
