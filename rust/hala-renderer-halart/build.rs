fn main() {
  // hala-renderer_amd/lib/libhalart.so is built by `python -c "import __graft_entry__ as g; g.build()"`
  let root = std::env::var("HALART_ROOT").unwrap_or_else(|_| "../..".to_string());
  println!("cargo:rustc-link-search=native={}/hala-renderer_amd/lib", root);
  println!("cargo:rustc-link-lib=dylib=halart");
}
