//! Nothing to export: the crate exists for its integration tests (tests/groups.rs, tests/domain.rs) -- the reference's own
//! test templates over the MI355X path.  See Cargo.toml and rust/ci.sh.
