// oracle/_ref driver: a C ABI over the REFERENCE's own octree storage class, compiled from the
// sources where they lie (`/root/reference/cpp/src/collision/detail/TreeNode.h` + `TreeNode.hxx`,
// which include nothing but the standard library).  This file holds NO reference code: it only
// instantiates `collision::detail::TreeNode<N>` and forwards calls.  Test infrastructure: it pins the
// oracle's dense-grid restatement of
//   rows (a)11  octree AND octree  (TreeNode::collides,      TreeNode.hxx:164-174, leaf :268)
//   rows (a)10  block storage half (set_block / union_block, TreeNode.hxx:74-95,140-148, leaf :255-259,267)
//   visit_leaves order             (TreeNode.hxx:176-190)   -- the order voxel sets are serialised in
// against the reference itself.  Built by `make -C oracle ref` into oracle/_ref/libref_treenode.so
// (git-ignored); never linked into the product.
#include <collision/detail/TreeNode.h>

#include <cstddef>
#include <cstdint>
#include <vector>

namespace {

using collision::detail::TreeNode;

struct TreeBase {
  virtual ~TreeBase() = default;
  virtual int      N() const = 0;
  virtual TreeBase *clone() const = 0;
  virtual uint64_t block(size_t bx, size_t by, size_t bz) const = 0;
  virtual void     set_block(size_t bx, size_t by, size_t bz, uint64_t v) = 0;
  virtual uint64_t union_block(size_t bx, size_t by, size_t bz, uint64_t v) = 0;
  virtual uint64_t intersect_block(size_t bx, size_t by, size_t bz, uint64_t v) = 0;
  virtual size_t   nblocks() const = 0;
  virtual int      is_empty() const = 0;
  virtual int      collides(const TreeBase &o) const = 0;
  virtual int      equals(const TreeBase &o) const = 0;
  virtual void     union_tree(const TreeBase &o) = 0;
  virtual void     intersect_tree(const TreeBase &o) = 0;
  virtual void     remove_tree(const TreeBase &o) = 0;
  virtual long     leaves(uint32_t *bxyz, uint64_t *vals, long cap) const = 0;
  virtual long     blocks_visited() const = 0;
};

template <size_t Nt> struct Tree final : TreeBase {
  TreeNode<Nt> t;
  const TreeNode<Nt> &other(const TreeBase &o) const { return static_cast<const Tree<Nt> &>(o).t; }
  int      N() const override { return int(Nt); }
  TreeBase *clone() const override { auto *c = new Tree<Nt>(); c->t = t; return c; }
  uint64_t block(size_t bx, size_t by, size_t bz) const override { return t.block(bx, by, bz); }
  void     set_block(size_t bx, size_t by, size_t bz, uint64_t v) override { t.set_block(bx, by, bz, v); }
  uint64_t union_block(size_t bx, size_t by, size_t bz, uint64_t v) override { return t.union_block(bx, by, bz, v); }
  uint64_t intersect_block(size_t bx, size_t by, size_t bz, uint64_t v) override { return t.intersect_block(bx, by, bz, v); }
  size_t   nblocks() const override { return t.nblocks(); }
  int      is_empty() const override { return t.is_empty(); }
  int      collides(const TreeBase &o) const override { return t.collides(other(o)); }
  int      equals(const TreeBase &o) const override { return t == other(o); }
  void     union_tree(const TreeBase &o) override { t.union_tree(other(o)); }
  void     intersect_tree(const TreeBase &o) override { t.intersect_tree(other(o)); }
  void     remove_tree(const TreeBase &o) override { t.remove_tree(other(o)); }
  long     leaves(uint32_t *bxyz, uint64_t *vals, long cap) const override {
    long n = 0;
    t.visit_leaves([&](size_t bx, size_t by, size_t bz, uint64_t v) {
      if (bxyz && vals && n < cap) {
        bxyz[3 * n] = uint32_t(bx); bxyz[3 * n + 1] = uint32_t(by); bxyz[3 * n + 2] = uint32_t(bz);
        vals[n] = v;
      }
      n++;
    });
    return n;
  }
  long     blocks_visited() const override {
    long n = 0;
    t.visit_blocks([&](size_t, size_t, size_t, uint64_t) { n++; });
    return n;
  }
};

}  // namespace

extern "C" {

// the sizes VoxelOctree's variant offers (collision/VoxelOctree.h) that matter here
void *ref_tree_new(int N) {
  switch (N) {
    case 8:   return new Tree<8>();
    case 16:  return new Tree<16>();
    case 32:  return new Tree<32>();
    case 64:  return new Tree<64>();
    case 128: return new Tree<128>();
    case 256: return new Tree<256>();
    case 512: return new Tree<512>();
    default:  return nullptr;
  }
}
void     ref_tree_free(void *t) { delete static_cast<TreeBase *>(t); }
void    *ref_tree_copy(const void *t) { return static_cast<const TreeBase *>(t)->clone(); }
int      ref_tree_N(const void *t) { return static_cast<const TreeBase *>(t)->N(); }
uint64_t ref_tree_block(const void *t, uint32_t bx, uint32_t by, uint32_t bz) { return static_cast<const TreeBase *>(t)->block(bx, by, bz); }
void     ref_tree_set_block(void *t, uint32_t bx, uint32_t by, uint32_t bz, uint64_t v) { static_cast<TreeBase *>(t)->set_block(bx, by, bz, v); }
uint64_t ref_tree_union_block(void *t, uint32_t bx, uint32_t by, uint32_t bz, uint64_t v) { return static_cast<TreeBase *>(t)->union_block(bx, by, bz, v); }
uint64_t ref_tree_intersect_block(void *t, uint32_t bx, uint32_t by, uint32_t bz, uint64_t v) { return static_cast<TreeBase *>(t)->intersect_block(bx, by, bz, v); }
uint64_t ref_tree_nblocks(const void *t) { return static_cast<const TreeBase *>(t)->nblocks(); }
int      ref_tree_is_empty(const void *t) { return static_cast<const TreeBase *>(t)->is_empty(); }
// both trees must have been created with the same N (the caller checks)
int      ref_tree_collides(const void *a, const void *b) { return static_cast<const TreeBase *>(a)->collides(*static_cast<const TreeBase *>(b)); }
int      ref_tree_equals(const void *a, const void *b) { return static_cast<const TreeBase *>(a)->equals(*static_cast<const TreeBase *>(b)); }
void     ref_tree_union_tree(void *a, const void *b) { static_cast<TreeBase *>(a)->union_tree(*static_cast<const TreeBase *>(b)); }
void     ref_tree_intersect_tree(void *a, const void *b) { static_cast<TreeBase *>(a)->intersect_tree(*static_cast<const TreeBase *>(b)); }
void     ref_tree_remove_tree(void *a, const void *b) { static_cast<TreeBase *>(a)->remove_tree(*static_cast<const TreeBase *>(b)); }
// visit_leaves in the reference's own traversal order; returns the number of leaves
long     ref_tree_leaves(const void *t, uint32_t *bxyz, uint64_t *vals, long cap) { return static_cast<const TreeBase *>(t)->leaves(bxyz, vals, cap); }
long     ref_tree_blocks_visited(const void *t) { return static_cast<const TreeBase *>(t)->blocks_visited(); }

// batch helpers so that Python drives millions of block operations without a call each
void ref_tree_union_blocks(void *t, const uint32_t *bxyz, const uint64_t *vals, long n, uint64_t *prev_out) {
  auto *tr = static_cast<TreeBase *>(t);
  for (long i = 0; i < n; i++) {
    uint64_t p = tr->union_block(bxyz[3 * i], bxyz[3 * i + 1], bxyz[3 * i + 2], vals[i]);
    if (prev_out) prev_out[i] = p;
  }
}
void ref_tree_set_blocks(void *t, const uint32_t *bxyz, const uint64_t *vals, long n) {
  auto *tr = static_cast<TreeBase *>(t);
  for (long i = 0; i < n; i++) tr->set_block(bxyz[3 * i], bxyz[3 * i + 1], bxyz[3 * i + 2], vals[i]);
}

}  // extern "C"
