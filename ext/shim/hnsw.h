/*
 * ext/shim/hnsw.h -- NOT pgvector's src/hnsw.h.  The fields, macros and functions of it that ext/hnswscan_gpu.c
 * touches, declared for the syntax check only (names as in src/hnsw.h:36-70, 182-222, 348-427, 470-475; inside the
 * extension the real header is used and HnswScanOpaqueData gains ONE field, marked "added").
 */
#ifndef EXT_SHIM_HNSW_H
#define EXT_SHIM_HNSW_H
#include "pgshim.h"
#include "ivfflat.h"			/* Vector */

#define HNSW_MAX_DIM 2000
#define HNSW_METAPAGE_BLKNO 0
#define HNSW_HEAD_BLKNO 1
#define HNSW_ELEMENT_TUPLE_TYPE 1
#define HNSW_NEIGHBOR_TUPLE_TYPE 2
#define HNSW_HEAPTIDS 10
#define HNSW_MAX_EF_SEARCH 1000
#define HNSW_DISTANCE_PROC 1
#define HNSW_NORM_PROC 2

extern int	hnsw_ef_search;
extern int	hnsw_iterative_scan;
#define HNSW_ITERATIVE_SCAN_OFF 0	/* src/hnsw.h: typedef enum HnswIterativeScanMode, first member */

typedef struct HnswMetaPageData
{
	uint32		magicNumber;
	uint32		version;
	uint32		dimensions;
	uint16		m;
	uint16		efConstruction;
	BlockNumber entryBlkno;
	OffsetNumber entryOffno;
	int16		entryLevel;
	BlockNumber insertPage;
}			HnswMetaPageData;
#define HnswPageGetMeta(page) ((HnswMetaPageData *) PageGetContents(page))

typedef struct HnswPageOpaqueData
{
	BlockNumber nextblkno;
	uint16		unused;
	uint16		page_id;
}			HnswPageOpaqueData;
typedef HnswPageOpaqueData *HnswPageOpaque;
#define HnswPageGetOpaque(page) ((HnswPageOpaque) PageGetSpecialPointer(page))

typedef struct HnswElementTupleData
{
	uint8		type;
	uint8		level;
	uint8		deleted;
	uint8		version;
	ItemPointerData heaptids[HNSW_HEAPTIDS];
	ItemPointerData neighbortid;
	uint16		unused;
	Vector		data;
}			HnswElementTupleData;
typedef HnswElementTupleData *HnswElementTuple;
#define HnswIsElementTuple(tup) ((tup)->type == HNSW_ELEMENT_TUPLE_TYPE)

typedef struct HnswNeighborTupleData
{
	uint8		type;
	uint8		version;
	uint16		count;
	ItemPointerData indextids[];
}			HnswNeighborTupleData;
typedef HnswNeighborTupleData *HnswNeighborTuple;

typedef struct HnswElementData HnswElementData;
typedef struct HnswNeighborArray HnswNeighborArray;
/* src/hnsw.h:171-180: pointers that are absolute in a serial build and relative (relptr) in a parallel one; the glue
 * only ever goes through the macros below, here in their absolute form */
typedef union
{
	HnswElementData *ptr;
}			HnswElementPtr;
typedef union
{
	HnswNeighborArray *ptr;
}			HnswNeighborArrayPtr;
typedef union
{
	HnswNeighborArrayPtr *ptr;
}			HnswNeighborsPtr;
typedef union
{
	char	   *ptr;
}			DatumPtr;
#define HnswPtrStore(base, hp, value) ((void) (base), (void) ((hp).ptr = (value)))
#define HnswPtrAccess(base, hp) ((void) (base), (hp).ptr)
#define HnswPtrIsNull(base, hp) ((void) (base), (hp).ptr == NULL)
#define HnswGetNeighbors(base, element, lc) HnswPtrAccess(base, HnswPtrAccess(base, (element)->neighbors)[lc])
#define HnswGetLayerM(m, layer) ((layer) == 0 ? (m) * 2 : (m))

struct HnswElementData
{
	HnswElementPtr next;
	ItemPointerData heaptids[HNSW_HEAPTIDS];
	uint8		heaptidsLength;
	uint8		level;
	uint8		deleted;
	uint8		version;
	uint32		hash;
	HnswNeighborsPtr neighbors;
	BlockNumber blkno;
	OffsetNumber offno;
	OffsetNumber neighborOffno;
	BlockNumber neighborPage;
	DatumPtr	value;
	LWLock		lock;
};
typedef HnswElementData *HnswElement;

typedef struct HnswCandidate
{
	HnswElementPtr element;
	float		distance;
	bool		closer;
}			HnswCandidate;

struct HnswNeighborArray
{
	int			length;
	bool		closerSet;
	HnswCandidate items[];
};

typedef struct HnswSearchCandidate
{
	HnswElementPtr element;
	double		distance;
}			HnswSearchCandidate;

typedef struct HnswTypeInfo
{
	int			maxDimensions;
}			HnswTypeInfo;

typedef struct HnswSupport
{
	FmgrInfo   *procinfo;
	FmgrInfo   *normprocinfo;
	Oid			collation;
}			HnswSupport;

typedef struct HnswScanOpaqueData
{
	const HnswTypeInfo *typeInfo;
	bool		first;
	List	   *w;
	int			m;
	int64		tuples;
	HnswSupport support;
	void	   *gpu;			/* added: PgvHnswScan of ext/hnswscan_gpu.c, NULL when the scan stays on the CPU */
}			HnswScanOpaqueData;
typedef HnswScanOpaqueData *HnswScanOpaque;

/* the in-memory phase of the build (src/hnsw.h:212-345): what ext/hnswbuild_gpu.c touches of it */
typedef struct HnswGraph
{
	HnswElementPtr head;
	double		indtuples;
	HnswElementPtr entryPoint;
	Size		memoryUsed;
	Size		memoryTotal;
	bool		flushed;
}			HnswGraph;

typedef struct HnswAllocator
{
	void	   *(*alloc) (Size size, void *state);
	void	   *state;
}			HnswAllocator;

typedef struct HnswBuildState
{
	Relation	heap;
	Relation	index;
	const HnswTypeInfo *typeInfo;
	int			dimensions;
	int			m;
	int			efConstruction;
	double		indtuples;
	HnswSupport support;
	HnswGraph	graphData;
	HnswGraph  *graph;
	double		ml;
	int			maxLevel;
	MemoryContext graphCtx;
	MemoryContext tmpCtx;
	HnswAllocator allocator;
	char	   *hnswarea;		/* NULL in a serial build; the shared area of a parallel one */
	void	   *gpu;			/* added: PgvHnswBuild of ext/hnswbuild_gpu.c, NULL when the build stays on the CPU */
}			HnswBuildState;

void	   *HnswAlloc(HnswAllocator * allocator, Size size);
HnswElement HnswInitElement(char *base, ItemPointer heaptid, int m, double ml, int maxLevel, HnswAllocator * allocator);
HnswElement HnswInitElementFromBlock(BlockNumber blkno, OffsetNumber offno);
void		HnswAddHeapTid(HnswElement element, ItemPointer heaptid);
FmgrInfo   *HnswOptionalProcInfo(Relation index, uint16 procnum);
const HnswTypeInfo *HnswGetTypeInfo(Relation index);
#endif
