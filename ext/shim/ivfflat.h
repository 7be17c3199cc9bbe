/*
 * ext/shim/ivfflat.h -- NOT pgvector's src/ivfflat.h.  The fields and functions of it that the glue in
 * ext/ touches, declared for the syntax check only (names as in src/ivfflat.h:188-314; inside the
 * extension the real header is used and gains ONE field per struct, marked "added").
 */
#ifndef EXT_SHIM_IVFFLAT_H
#define EXT_SHIM_IVFFLAT_H
#include "pgshim.h"

#define IVFFLAT_MAX_DIM 2000
#define IVFFLAT_HEAD_BLKNO 1
#define IVFFLAT_DISTANCE_PROC 1
#define IVFFLAT_NORM_PROC 2
#define IVFFLAT_KMEANS_NORM_PROC 4

typedef struct Vector
{
	int32		vl_len_;
	int16		dim;
	int16		unused;
	float		x[];
}			Vector;

typedef struct IvfflatPageOpaqueData
{
	BlockNumber nextblkno;
	uint16		unused;
	uint16		page_id;
}			IvfflatPageOpaqueData;
typedef IvfflatPageOpaqueData *IvfflatPageOpaque;
#define IvfflatPageGetOpaque(page) ((IvfflatPageOpaque) PageGetSpecialPointer(page))

typedef struct IvfflatListData
{
	BlockNumber startPage;
	BlockNumber insertPage;
	Vector		center;
}			IvfflatListData;
typedef IvfflatListData *IvfflatList;

typedef struct IvfflatTypeInfo
{
	int			maxDimensions;
	Size		(*itemSize) (int dimensions);
}			IvfflatTypeInfo;

typedef struct VectorArrayData
{
	int			length;
	int			maxlen;
	int			dim;
	Size		itemsize;
	char	   *items;
}			VectorArrayData;
typedef VectorArrayData *VectorArray;
#define VectorArrayGet(arr, offset) (((char *) (arr)->items) + (offset) * (arr)->itemsize)

typedef struct IvfflatScanOpaqueData
{
	const IvfflatTypeInfo *typeInfo;
	int			probes;
	int			maxProbes;
	int			dimensions;
	bool		first;
	Datum		value;
	FmgrInfo   *normprocinfo;
	BufferAccessStrategy bas;
	int			listIndex;
	void	   *gpu;			/* added: PgvIvfScan of ext/ivfscan_gpu.c, NULL when the scan stays on the CPU */
}			IvfflatScanOpaqueData;
typedef IvfflatScanOpaqueData *IvfflatScanOpaque;

typedef struct IvfflatBuildState
{
	Relation	index;
	const IvfflatTypeInfo *typeInfo;
	int			dimensions;
	int			lists;
	VectorArray samples;
	VectorArray centers;
	FmgrInfo   *kmeansnormprocinfo;
	void	   *gpu;			/* added: PgvIvfBuild of ext/ivfbuild_gpu.c */
}			IvfflatBuildState;

void		IvfflatGetMetaPageInfo(Relation index, int *lists, int *dimensions);
const IvfflatTypeInfo *IvfflatGetTypeInfo(Relation index);
FmgrInfo   *IvfflatOptionalProcInfo(Relation index, uint16 procnum);
double		RandomDouble(void);
int			RandomInt(void);
#endif
