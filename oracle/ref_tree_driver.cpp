// Driver around the REFERENCE's own Newick importer (compiled from /root/reference/src/tree.cpp
// where it lies; see oracle/Makefile).  Test infrastructure: pins the import order of a-11
// (leaf ids in order of appearance, internal ids totalLeaves, totalLeaves+1, ... in order of '(').
#include "tree.hpp"
#include <cstring>
#include <vector>

extern "C" {

// Parses `newick` with the reference's Tree(newick,totalLeaves) and flattens it in pre-order.
// Returns the number of nodes (or -needed if cap is too small).  names: cap x 64 bytes.
__attribute__((visibility("default")))
int ref_tree_flatten(const char* newick, long totalLeaves, int cap, int* idx, int* parent_idx,
                     double* bl, int* is_leaf, char* names)
{
    Tree t(std::string(newick), (size_t)totalLeaves);
    std::vector<Node*> order, stack;
    stack.push_back(t.root);
    while (!stack.empty()) {
        Node* n = stack.back(); stack.pop_back();
        order.push_back(n);
        for (size_t i = n->children.size(); i-- > 0;) stack.push_back(n->children[i]);
    }
    if ((int)order.size() > cap) return -(int)order.size();
    for (size_t i = 0; i < order.size(); ++i) {
        Node* n = order[i];
        idx[i] = n->idx;
        parent_idx[i] = n->parent ? n->parent->idx : -1;
        bl[i] = n->bl;
        is_leaf[i] = n->children.empty() ? 1 : 0;
        std::strncpy(names + 64 * i, n->name.c_str(), 63);
        names[64 * i + 63] = 0;
    }
    return (int)order.size();
}

}
